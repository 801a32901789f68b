"""Tensor-level front end of the C-ABI kernels (one function per entry point).

PyTorch is plumbing here: it owns device memory and the current HIP stream; all arithmetic
happens in libdeep3d_planesweep.so.  Tensors must be fp32, contiguous and on the GPU --
anything else raises (no CPU path exists).  Shapes are unbatched, as in the header.
"""
import collections
import ctypes
import os

import torch

from . import _lib
from . import config as _cfg

PER_PLANE, PER_PIXEL, AFFINE = 0, 1, 2


class AffineDepth:
    """Per-pixel depth hypotheses in their generating form (include/deep3d_planesweep.h, D3D_DEPTH_AFFINE): `maps` is
    [2,h,w] = (lo, step) and plane k of a pixel lies at lo + k * step, k < D -- what module.py:616-631 builds its
    [D,h,w] volume from.  The aggregation and regression ops (variance_volume*, weighted_corr, pair_corr_mean, softargmin_conf4*)
    take it in place of the volume and read two maps instead of D; homo_warp and pair_softmax_max take [D] or [D,h,w] only
    (pass `.volume()`)."""

    def __init__(self, maps, D):
        if not (isinstance(maps, torch.Tensor) and maps.dim() == 3 and maps.shape[0] == 2):
            raise ValueError("AffineDepth maps must be [2,h,w]")
        self.maps, self.D = maps, int(D)

    def volume(self):
        """The [D,h,w] volume these maps stand for (same two roundings as the kernels': product, then sum)."""
        k = torch.arange(self.D, dtype=torch.float32, device=self.maps.device).view(-1, 1, 1)
        return (self.maps[0:1] + k * self.maps[1:2]).contiguous()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The HIP stream torch currently queues work on.  torch.cuda.current_stream() builds a Stream object through several
    Python layers (8 us; an AdaMVS view makes 860 launches: tools/host_profile.py); the raw handle is one C call."""
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# Which kernels served the calls so far: name -> count.  The model-level parity tests clear it, run a forward and assert that
# the production kernels (tile convolutions, fused conv-GRU cell, channel-last volumes, window / ring sweeps) were the ones
# dispatched -- not a fallback that happens to give the same numbers.
dispatch_counts = collections.Counter()
CONV2D_ZS_MINPIX = 256 * 256   # smallest image (pixels) conv2d_k3 hands to the 2-D tile kernel


def sweep_dispatch_counts(reset=False):
    """{'direct': n, 'tiled': n, 'window': n}: sweep calls served by each kernel family (the C dispatcher's own counters)."""
    buf = (ctypes.c_ulonglong * 4)()
    _lib.check(_lib.load().d3d_debug_dispatch_counts(buf, int(bool(reset))), "d3d_debug_dispatch_counts")
    return {"direct": int(buf[1]), "tiled": int(buf[2]), "window": int(buf[3])}


def _chk(t, name, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s is on %s: the plane-sweep engine only runs on the GPU (no CPU fallback)"
                           % (name, t.device))
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (got %s)" % (name, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must have %d dims (got shape %s)" % (name, ndim, tuple(t.shape)))
    return ctypes.c_void_p(t.data_ptr())


_forced = [None]


def _sync_force_path():
    """D3D_FORCE_PATH = direct | tiled (tests, profiling): forwarded to the library's test hook when it changes."""
    want = _cfg.get("D3D_FORCE_PATH")
    if want != _forced[0]:
        code = {"": 0, "auto": 0, "direct": 1, "tiled": 2, "window": 3}.get(want)
        if code is None:
            raise ValueError("D3D_FORCE_PATH must be direct, tiled, window or unset (got %r)" % want)
        _lib.check(_lib.load().d3d_debug_force_path(code), "d3d_debug_force_path")
        _forced[0] = want


def _workspace(n_views, C, D, h, w, elem_bytes, device, mode=PER_PIXEL):
    """Scratch for one sweep call, sized by the library FOR THE CALL'S DEPTH MODE (the window kernel's channel-last copy -- 650 MB
    at the last cascade stage -- serves hypothesis volumes only: (lo, step) maps and per-plane depths do not ask for it) and owned
    by torch's caching allocator: the allocator hands the block back only after the work queued on the current stream (this call)
    has been ordered, so calls never share it."""
    _sync_force_path()
    n = int(_lib.load().d3d_sweep_workspace_bytes_for(n_views, C, D, h, w, elem_bytes, mode))
    if n == 0:
        return None, ctypes.c_void_p(0), 0
    buf = torch.empty((n,), dtype=torch.uint8, device=device)
    return buf, ctypes.c_void_p(buf.data_ptr()), n


def _opt(t, name):
    return None if t is None else _chk(t, name)


def _depth(depth, h, w, affine_ok=True, op=""):
    if isinstance(depth, AffineDepth):
        if not affine_ok:
            raise TypeError("%s takes depth hypotheses as [D] or [D,h,w]; pass AffineDepth.volume()" % op)
        if tuple(depth.maps.shape[1:]) != (h, w):
            raise ValueError("affine depth maps must be [2,%d,%d] (got %s)" % (h, w, tuple(depth.maps.shape)))
        return _chk(depth.maps, "depth.maps", 3), AFFINE, depth.D
    if depth.dim() == 1:
        return _chk(depth, "depth"), PER_PLANE, depth.shape[0]
    if depth.dim() == 3:
        if tuple(depth.shape[1:]) != (h, w):
            raise ValueError("per-pixel depth must be [D,%d,%d] (got %s)" % (h, w, tuple(depth.shape)))
        return _chk(depth, "depth"), PER_PIXEL, depth.shape[0]
    raise ValueError("depth must be [D] or [D,h,w] (got %s)" % (tuple(depth.shape),))


def _ptr_array(tensors, name):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if _chk(t, "%s[%d]" % (name, i), 3) is not None else None
    return arr


def compose_projections(proj44):
    """[V,4,4] (index 0 = reference) -> [V-1,12] composed [rot|trans] (module.py:528-530)."""
    p = _chk(proj44, "proj44", 3)
    V = proj44.shape[0]
    if tuple(proj44.shape[1:]) != (4, 4):
        raise ValueError("proj44 must be [V,4,4]")
    out = torch.empty((V - 1, 12), dtype=torch.float32, device=proj44.device)
    _lib.check(_lib.load().d3d_compose_projections(p, V, _chk(out, "out"), _stream()), "d3d_compose_projections")
    return out


def homo_warp(src, proj34, depth, out=None):
    """src [C,h,w], proj34 [12], depth [D]|[D,h,w] -> [C,D,h,w]."""
    C, h, w = src.shape
    dp, mode, D = _depth(depth, h, w, affine_ok=False, op="homo_warp")
    if out is None:
        out = torch.empty((C, D, h, w), dtype=torch.float32, device=src.device)
    ws, wp, wn = _workspace(2, C, D, h, w, 4, src.device, mode)
    rc = _lib.load().d3d_homo_warp(_chk(src, "src", 3), _chk(proj34, "proj34"), dp, mode, C, D, h, w,
                                   _chk(out, "out", 4), wp, wn, _stream())
    _lib.check(rc, "d3d_homo_warp")
    return out


def homo_warp_double(src, src_proj, ref_proj, depth):
    """module.py:560-601 homo_warping_double: src [C,h,w] fp32, src_proj / ref_proj [4,4] FLOAT64 (the reference function
    only accepts double matrices), depth [D]|[D,h,w] fp32 -> [C,D,h,w] fp32 sampled at fp64-computed coordinates."""
    for name, t in (("src_proj", src_proj), ("ref_proj", ref_proj)):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and tuple(t.shape) == (4, 4)):
            raise TypeError("%s must be a CUDA float64 [4,4] tensor" % name)
    C, h, w = src.shape
    dp, mode, D = _depth(depth, h, w, affine_ok=False, op="homo_warp_double")
    p44 = torch.stack([ref_proj, src_proj]).contiguous()
    p34 = torch.empty((1, 12), dtype=torch.float64, device=src.device)
    lib = _lib.load()
    _lib.check(lib.d3d_compose_projections_f64(ctypes.c_void_p(p44.data_ptr()), 2, ctypes.c_void_p(p34.data_ptr()),
                                               _stream()), "d3d_compose_projections_f64")
    out = torch.empty((C, D, h, w), dtype=torch.float32, device=src.device)
    rc = lib.d3d_homo_warp_f64coord(_chk(src, "src", 3), ctypes.c_void_p(p34.data_ptr()), dp, mode, C, D, h, w,
                                    _chk(out, "out", 4), _stream())
    _lib.check(rc, "d3d_homo_warp_f64coord")
    return out


def _check_feats(feats, proj34):
    if len(feats) < 2:
        raise ValueError("need a reference and at least one source view")
    shape = tuple(feats[0].shape)
    for f in feats:
        if tuple(f.shape) != shape:
            raise ValueError("all feature maps must share one shape (got %s vs %s)" % (tuple(f.shape), shape))
    if proj34.numel() != 12 * (len(feats) - 1):
        raise ValueError("proj34 must hold %d x 12 floats" % (len(feats) - 1))
    return shape


def _chk16(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float16 and t.is_contiguous()):
        raise TypeError("%s must be a contiguous CUDA float16 tensor" % name)
    return ctypes.c_void_p(t.data_ptr())


def variance_volume(feats, proj34, depth, out=None, plane_major=False):
    """feats: list of V tensors [C,h,w] (feats[0] = reference); -> [C,D,h,w] (cas_mvsnet.py:45-60), or [D,C,h,w] with
    plane_major=True (fp32: plane d is one contiguous block for the slice loops).
    float16 feature maps give a float16 volume (fp32 arithmetic, BASELINE config 5)."""
    C, h, w = _check_feats(feats, proj34)
    if plane_major and feats[0].dtype == torch.float32:
        dp, mode, D = _depth(depth, h, w)
        if out is None:
            out = torch.empty((D, C, h, w), dtype=torch.float32, device=feats[0].device)
        arr = _ptr_array(feats, "feats")
        ws, wp, wn = _workspace(len(feats), C, D, h, w, 4, feats[0].device, mode)
        rc = _lib.load().d3d_variance_volume_planes(arr, _chk(proj34, "proj34"), dp, mode, len(feats), C, D, h, w,
                                                    _chk(out, "out", 4), wp, wn, _stream())
        _lib.check(rc, "d3d_variance_volume_planes")
        return out
    if plane_major:
        raise ValueError("plane_major volumes are fp32")
    if feats[0].dtype == torch.float16:
        dp, mode, D = _depth(depth, h, w)
        if out is None:
            out = torch.empty((C, D, h, w), dtype=torch.float16, device=feats[0].device)
        arr = (ctypes.c_void_p * len(feats))(*[_chk16(f, "feats[%d]" % i).value for i, f in enumerate(feats)])
        ws, wp, wn = _workspace(len(feats), C, D, h, w, 2, feats[0].device, mode)
        rc = _lib.load().d3d_variance_volume_f16(arr, _chk(proj34, "proj34"), dp, mode, len(feats), C, D, h, w,
                                                 _chk16(out, "out"), wp, wn, _stream())
        _lib.check(rc, "d3d_variance_volume_f16")
        return out
    dp, mode, D = _depth(depth, h, w)
    if out is None:
        out = torch.empty((C, D, h, w), dtype=torch.float32, device=feats[0].device)
    arr = _ptr_array(feats, "feats")
    ws, wp, wn = _workspace(len(feats), C, D, h, w, 4, feats[0].device, mode)
    rc = _lib.load().d3d_variance_volume(arr, _chk(proj34, "proj34"), dp, mode, len(feats), C, D, h, w,
                                         _chk(out, "out", 4), wp, wn, _stream())
    _lib.check(rc, "d3d_variance_volume")
    return out


def variance_volume_cl(feats, proj34, depth, layout="cl"):
    """variance_volume for the bf16 mode of the regulariser (BASELINE config 3): fp32 features and arithmetic, the volume
    written once as a channel-last bf16 tensor [D,h,w,C] (the rounding conv0 would apply when it stages the planar volume),
    or with layout="cl8" in planes of 8-channel groups [D,C/8,h,w,8] -- whole 16-byte cells per store, the form
    `conv3d_k3_cl` takes as a 5-d tensor (for C = 8 the same bytes).  Shapes the ring / window kernels do not take go
    through the planar kernel and the format conversion."""
    if layout not in ("cl", "cl8"):
        raise ValueError("layout must be 'cl' or 'cl8'")
    C, h, w = _check_feats(feats, proj34)
    dp, mode, D = _depth(depth, h, w)
    if C % 8 == 0:
        out = torch.empty((D, C // 8, h, w, 8) if layout == "cl8" else (D, h, w, C), dtype=h16_dtype(), device=feats[0].device)
        arr = _ptr_array(feats, "feats")
        ws, wp, wn = _workspace(len(feats), C, D, h, w, 4, feats[0].device, mode)
        name = "d3d_variance_volume_cl8_h16" if layout == "cl8" else "d3d_variance_volume_cl_h16"
        rc = getattr(_lib.load(), name)(arr, _chk(proj34, "proj34"), dp, mode, len(feats), C, D, h, w,
                                        ctypes.c_void_p(out.data_ptr()), wp, wn, _stream())
        if rc != _lib.ERR_UNSUPPORTED:
            _lib.check(rc, name)
            dispatch_counts["variance_" + layout] += 1
            return out
    dispatch_counts["variance_cl_fallback"] += 1
    y = to_cl(variance_volume(feats, proj34, depth))
    return cl_to_cl8(y) if layout == "cl8" else y


def cl_to_cl8(x):
    """[D,H,W,C] -> [D,C/8,H,W,8] (torch permutation: the fallback route of variance_volume_cl and the tests' reference)."""
    D, H, W, C = x.shape
    return x.view(D, H, W, C // 8, 8).permute(0, 3, 1, 2, 4).contiguous()


def cl8_to_cl(x):
    D, G, H, W, _ = x.shape
    return x.permute(0, 2, 3, 1, 4).reshape(D, H, W, G * 8).contiguous()


def weighted_corr(feats, proj34, weights, depth, out=None, plane_major=False):
    """adamvs.py:492-509. weights [V-1,h,w] -> [C,D,h,w], or [D,C,h,w] with plane_major=True (plane d contiguous)."""
    C, h, w = _check_feats(feats, proj34)
    if tuple(weights.shape) != (len(feats) - 1, h, w):
        raise ValueError("weights must be [%d,%d,%d]" % (len(feats) - 1, h, w))
    dp, mode, D = _depth(depth, h, w)
    if out is None:
        out = torch.empty((D, C, h, w) if plane_major else (C, D, h, w), dtype=torch.float32, device=feats[0].device)
    arr = _ptr_array(feats, "feats")
    ws, wp, wn = _workspace(len(feats), C, D, h, w, 4, feats[0].device, mode)
    rc = _lib.load().d3d_weighted_corr(arr, _chk(proj34, "proj34"), _chk(weights, "weights", 3), dp, mode,
                                       len(feats), C, D, h, w, int(bool(plane_major)), _chk(out, "out", 4), wp, wn, _stream())
    _lib.check(rc, "d3d_weighted_corr")
    return out


def weighted_corr_cl8(feats, proj34, weights, depth, out=None):
    """weighted_corr for the fast mode of the slice regularisers: the volume leaves the sweep as 16-bit cells in planes of
    8-channel groups, [D, C/8, h, w, 8] in the library's h16 format (d3d_weighted_corr_cl8_h16) -- plane d is what the fused
    conv-GRU cell stages with 16-byte loads (gru_cell_conv_fused on a 4-d cost).  The values are the fp32 volume's, rounded
    once (the rounding the cell applies to the planar plane anyway).  None where the window kernel does not take the shape or
    D3D_KERNELS_OFF=corr_cl8: the caller then takes weighted_corr(plane_major=True)."""
    C, h, w = _check_feats(feats, proj34)
    if tuple(weights.shape) != (len(feats) - 1, h, w):
        raise ValueError("weights must be [%d,%d,%d]" % (len(feats) - 1, h, w))
    if C % 8 or _cfg.off("corr_cl8"):
        return None
    dp, mode, D = _depth(depth, h, w)
    shape = (D, C // 8, h, w, 8)
    if out is None:
        out = torch.empty(shape, dtype=h16_dtype(), device=feats[0].device)
    elif tuple(out.shape) != shape or out.dtype != h16_dtype() or not out.is_contiguous():
        raise ValueError("out must be a contiguous %s tensor of ops.h16_dtype()" % (shape,))
    arr = _ptr_array(feats, "feats")
    ws, wp, wn = _workspace(len(feats), C, D, h, w, 4, feats[0].device, mode)
    rc = _lib.load().d3d_weighted_corr_cl8_h16(arr, _chk(proj34, "proj34"), _chk(weights, "weights", 3), dp, mode, len(feats), C, D, h, w,
                                               ctypes.c_void_p(out.data_ptr()), wp, wn, _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "d3d_weighted_corr_cl8_h16")
    dispatch_counts["weighted_corr_cl8"] += 1
    return out


def pair_corr_mean(ref, src, proj34, depth, out=None):
    """adamvs.py:469-474. -> [D,h,w]."""
    C, h, w = _check_feats([ref, src], proj34)
    dp, mode, D = _depth(depth, h, w)
    if out is None:
        out = torch.empty((D, h, w), dtype=torch.float32, device=ref.device)
    ws, wp, wn = _workspace(2, C, D, h, w, 4, ref.device, mode)
    rc = _lib.load().d3d_pair_corr_mean(_chk(ref, "ref", 3), _chk(src, "src", 3), _chk(proj34, "proj34"), dp, mode,
                                        C, D, h, w, _chk(out, "out", 3), wp, wn, _stream())
    _lib.check(rc, "d3d_pair_corr_mean")
    return out


def softargmin_conf4(cost, depth):
    """cost [D,h,w] -> (depth [h,w], confidence [h,w]) (cas_mvsnet.py:69-76)."""
    D, h, w = cost.shape
    dp, mode, D2 = _depth(depth, h, w)
    if D2 != D:
        raise ValueError("depth has %d planes, cost has %d" % (D2, D))
    dep = torch.empty((h, w), dtype=torch.float32, device=cost.device)
    conf = torch.empty_like(dep)
    rc = _lib.load().d3d_softargmin_conf4(_chk(cost, "cost", 3), dp, mode, D, h, w, _chk(dep, "dep"),
                                          _chk(conf, "conf"), _stream())
    _lib.check(rc, "d3d_softargmin_conf4")
    return dep, conf


def softargmin_conf4_var(cost, depth, lamb):
    """cost [D,h,w] -> (depth, confidence, exp_variance) each [h,w] (ucsnet.py:137-151)."""
    D, h, w = cost.shape
    dp, mode, D2 = _depth(depth, h, w)
    if D2 != D:
        raise ValueError("depth has %d planes, cost has %d" % (D2, D))
    dep = torch.empty((h, w), dtype=torch.float32, device=cost.device)
    conf, var = torch.empty_like(dep), torch.empty_like(dep)
    rc = _lib.load().d3d_softargmin_conf4_var(_chk(cost, "cost", 3), dp, mode, D, h, w, float(lamb), _chk(dep, "dep"),
                                              _chk(conf, "conf"), _chk(var, "var"), _stream())
    _lib.check(rc, "d3d_softargmin_conf4_var")
    return dep, conf, var


def uncertainty_aware_samples(cur_depth, exp_var, ndepth, shape=None, affine=False):
    """ucsnet.py:30-53.  First stage: cur_depth [2+] = (min, ..., max) -> uniform hypotheses [ndepth] (the reference tiles
    them to [ndepth,h,w]; the sweep kernels take the per-plane form).  Later stages: cur_depth, exp_var [h,w] ->
    [ndepth,h,w] hypotheses between cur - var and cur + var -- or, affine=True, the two maps (low, step) that generate them as an
    AffineDepth: plane k = low + k * step with the kernel's two roundings; the reference's `+ 1e-12` is the identity on any
    depth above 2e-5 (half an ulp there), so for real scenes the planes are the volume's, bit for bit."""
    if cur_depth.dim() == 1:
        return depth_range_samples(cur_depth, ndepth, 0.0)
    if ndepth <= 1:
        raise ValueError("ndepth must be > 1")
    h, w = cur_depth.shape
    if exp_var is None or tuple(exp_var.shape) != (h, w):
        raise ValueError("exp_var must be [%d,%d]" % (h, w))
    if affine:
        maps = torch.empty((2, h, w), dtype=torch.float32, device=cur_depth.device)
        torch.sub(cur_depth, exp_var, out=maps[0])                                  # low
        torch.sub(cur_depth + exp_var, maps[0], out=maps[1])                        # high - low
        # step: the kernel's fp32 DIVISION (a tensor divisor: torch turns a scalar divisor into a multiplication by its reciprocal)
        maps[1].div_(torch.full((1,), float(ndepth) - 1.0, dtype=torch.float32, device=cur_depth.device))
        return AffineDepth(maps, ndepth)
    out = torch.empty((ndepth, h, w), dtype=torch.float32, device=cur_depth.device)
    rc = _lib.load().d3d_uncertainty_samples(_chk(cur_depth, "cur_depth", 2), _chk(exp_var, "exp_var", 2), ndepth, h, w,
                                             _chk(out, "out"), _stream())
    _lib.check(rc, "d3d_uncertainty_samples")
    return out


def pair_softmax_max(score, depth):
    """score [D,h,w] -> (view_weight [h,w], pair_depth [h,w]) (adamvs.py:478-486)."""
    D, h, w = score.shape
    dp, mode, D2 = _depth(depth, h, w, affine_ok=False, op="pair_softmax_max")
    if D2 != D:
        raise ValueError("depth has %d planes, score has %d" % (D2, D))
    vw = torch.empty((h, w), dtype=torch.float32, device=score.device)
    pd = torch.empty_like(vw)
    rc = _lib.load().d3d_pair_softmax_max(_chk(score, "score", 3), dp, mode, D, h, w, _chk(vw, "vw"),
                                          _chk(pd, "pd"), _stream())
    _lib.check(rc, "d3d_pair_softmax_max")
    return vw, pd


def online_regress_update(reg, dplane, max_p, sum_d, sum_p):
    """One plane of adamvs.py:514-525; reg/max_p/sum_d/sum_p [H,W], dplane [hd,wd] (resampled if smaller)."""
    H, W = reg.shape
    hd, wd = dplane.shape
    rc = _lib.load().d3d_online_regress_update(_chk(reg, "reg", 2), _chk(dplane, "dplane", 2), hd, wd, H, W,
                                               _chk(max_p, "max_p", 2), _chk(sum_d, "sum_d", 2),
                                               _chk(sum_p, "sum_p", 2), _stream())
    _lib.check(rc, "d3d_online_regress_update")


def slice_head_regress(up, weight, bias, transposed, dplane, max_p, sum_d, sum_p):
    """reg = upconv2d(up) + bias (ConvTranspose2d(8,1,3,2,1,1) if `transposed` else Conv2d(8,1,3,pad 1); adamvs.py:417-418) and
    the online regression update of that plane (adamvs.py:514-525) in ONE streaming kernel: `reg` never reaches memory.  bf16
    mode only (operands rounded to bf16 as the matrix cores round them).  Returns False when the fused kernel does not apply
    (the caller then runs the layer and online_regress_update)."""
    if conv_precision() != "h16" or _cfg.off("head_fused") or up.dim() != 3 or up.shape[0] != 8 or bias is None:
        return False
    _, h, w = up.shape
    H, W = (2 * h, 2 * w) if transposed else (h, w)
    if tuple(max_p.shape) != (H, W) or weight.numel() != 72:
        return False
    if w % (2 if transposed else 4):
        return False
    hd, wd = dplane.shape
    wr = derived_weight(weight, "h16round", lambda t: t.to(h16_dtype()).float().contiguous())   # the matrix cores' rounding, once
    rc = _lib.load().d3d_slice_head_regress_h16(_chk(up, "up", 3), _chk(wr, "weight"), _chk(bias, "bias"), int(bool(transposed)),
                                                 _chk(dplane, "dplane", 2), hd, wd, h, w, _chk(max_p, "max_p", 2),
                                                 _chk(sum_d, "sum_d", 2), _chk(sum_p, "sum_p", 2), _stream())
    _lib.check(rc, "d3d_slice_head_regress_h16")
    dispatch_counts["slice_head_regress"] += 1
    return True


def slice_tail_regress_same(state2, w_up, b_up, state1, skip_after_act, w_head, b_head, dplane, max_p, sum_d, sum_p):
    """upconv1 + skip -> Conv2d(8, 1, 3, pad 1) head -> online regression update in ONE kernel for the stages whose head keeps `up`'s
    resolution (adamvs.py:413-418 at the last stage: relu(upconv1 + bias + state1); msrednet.py:361-363: relu(upconv1) + state1 with
    skip_after_act): `up` and `reg` never reach memory (d3d_slice_tail_regress_same_h16; bit for bit what convtranspose2d_k3s2 +
    slice_head_regress give).  w_head: the [1,8,3,3] Conv2d weight.  h16 mode only; False when it does not apply."""
    if conv_precision() != "h16" or _cfg.off("tail_fused") or _cfg.off("head_fused") or _cfg.off("tail_same") or state2.dim() != 3 \
            or state2.shape[0] != 16 or b_head is None:
        return False
    _, h, w = state2.shape
    if tuple(state1.shape) != (8, 2 * h, 2 * w) or tuple(max_p.shape) != (2 * h, 2 * w) or w % 4 \
            or tuple(w_up.shape) != (16, 8, 3, 3) or tuple(w_head.shape) != (1, 8, 3, 3):
        return False
    hd, wd = dplane.shape
    wp = derived_weight(w_up, "t2dbf16", _pack_t2d_bf16)
    wr = derived_weight(w_head, "h16round", lambda t: t.to(h16_dtype()).float().contiguous())
    rc = _lib.load().d3d_slice_tail_regress_same_h16(_chk(state2, "state2", 3), ctypes.c_void_p(wp.data_ptr()), _opt(b_up, "b_up"),
                                                      _chk(state1, "state1", 3), int(bool(skip_after_act)), _chk(wr, "w_head"),
                                                      _chk(b_head, "b_head"), _chk(dplane, "dplane", 2), hd, wd, h, w,
                                                      _chk(max_p, "max_p", 2), _chk(sum_d, "sum_d", 2), _chk(sum_p, "sum_p", 2), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return False
    _lib.check(rc, "d3d_slice_tail_regress_same_h16")
    dispatch_counts["slice_tail_regress_same"] += 1
    return True


def slice_tail_regress(state2, w_up, b_up, state1, w_head, b_head, dplane, max_p, sum_d, sum_p):
    """relu(upconv1(state2) + state1) -> upconv2d -> online regression update (adamvs.py:413-418, 423-425, 514-525) in ONE kernel
    for the stages whose head is the stride-2 ConvTranspose2d: `up` and `reg` never reach memory (d3d_slice_tail_regress_h16; bit
    for bit what convtranspose2d_k3s2 + slice_head_regress give).  bf16 mode only; False when it does not apply."""
    if conv_precision() != "h16" or _cfg.off("tail_fused") or _cfg.off("head_fused") or state2.dim() != 3 or state2.shape[0] != 16 \
            or b_up is None or b_head is None:
        return False
    _, h, w = state2.shape
    if tuple(state1.shape) != (8, 2 * h, 2 * w) or tuple(max_p.shape) != (4 * h, 4 * w) or w % 4 \
            or tuple(w_up.shape) != (16, 8, 3, 3) or tuple(w_head.shape) != (8, 1, 3, 3):
        return False
    hd, wd = dplane.shape
    wp = derived_weight(w_up, "t2dbf16", _pack_t2d_bf16)
    wr = derived_weight(w_head, "h16round", lambda t: t.to(h16_dtype()).float().contiguous())
    rc = _lib.load().d3d_slice_tail_regress_h16(_chk(state2, "state2", 3), ctypes.c_void_p(wp.data_ptr()), _chk(b_up, "b_up"),
                                                 _chk(state1, "state1", 3), _chk(wr, "w_head"), _chk(b_head, "b_head"),
                                                 _chk(dplane, "dplane", 2), hd, wd, h, w, _chk(max_p, "max_p", 2),
                                                 _chk(sum_d, "sum_d", 2), _chk(sum_p, "sum_p", 2), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return False
    _lib.check(rc, "d3d_slice_tail_regress_h16")
    dispatch_counts["slice_tail_regress"] += 1
    return True


def online_regress_finalize(max_p, sum_d, sum_p):
    dep = torch.empty_like(sum_d)
    conf = torch.empty_like(sum_d)
    rc = _lib.load().d3d_online_regress_finalize(_chk(max_p, "max_p"), _chk(sum_d, "sum_d"), _chk(sum_p, "sum_p"),
                                                 sum_d.numel(), _chk(dep, "dep"), _chk(conf, "conf"), _stream())
    _lib.check(rc, "d3d_online_regress_finalize")
    return dep, conf


def depth_range_samples(cur_depth, D, interval, h=0, w=0):
    """cur_depth [2] -> [D]; cur_depth [h,w] -> [D,h,w] (module.py:616-650)."""
    if cur_depth.dim() == 1:
        if cur_depth.numel() < 2:
            raise ValueError("cur_depth must hold (min, ..., max)")
        mm = torch.stack([cur_depth[0], cur_depth[-1]]).contiguous()
        out = torch.empty((D,), dtype=torch.float32, device=cur_depth.device)
        rc = _lib.load().d3d_depth_range_samples(_chk(mm, "cur_depth"), PER_PLANE, D, 0.0, 0, 0, _chk(out, "out"),
                                                 _stream())
    else:
        h, w = cur_depth.shape
        out = torch.empty((D, h, w), dtype=torch.float32, device=cur_depth.device)
        rc = _lib.load().d3d_depth_range_samples(_chk(cur_depth, "cur_depth", 2), PER_PIXEL, D, float(interval), h,
                                                 w, _chk(out, "out"), _stream())
    _lib.check(rc, "d3d_depth_range_samples")
    return out


def depth_range_affine(cur_depth, D, interval):
    """cur_depth [h,w] -> AffineDepth([2,h,w] = (lo, step), D): the two maps module.py:616-631 generates its D planes
    from (depth_range_samples(cur_depth, D, interval) == depth_range_affine(cur_depth, D, interval).volume(), bit for bit)."""
    h, w = cur_depth.shape
    out = torch.empty((2, h, w), dtype=torch.float32, device=cur_depth.device)
    rc = _lib.load().d3d_depth_range_samples(_chk(cur_depth, "cur_depth", 2), AFFINE, D, float(interval), h, w,
                                             _chk(out, "out"), _stream())
    _lib.check(rc, "d3d_depth_range_samples")
    return AffineDepth(out, D)


def resize_bilinear(x, H, W):
    """[n,h,w] -> [n,H,W], F.interpolate(bilinear, align_corners=False) semantics."""
    n, h, w = x.shape
    out = torch.empty((n, H, W), dtype=torch.float32, device=x.device)
    rc = _lib.load().d3d_resize_bilinear(_chk(x, "x", 3), n, h, w, H, W, _chk(out, "out"), _stream())
    _lib.check(rc, "d3d_resize_bilinear")
    return out


def _pack_c8_bf16(w, dt=None):
    """[8,Ci,3,3,3] -> the B operands of v_mfma_f32_16x16x32_bf16 for d3d_conv3d_k3_c8_h16: [kz][K block][lane][8] bf16
    with K = (ky, kx, ci) padded to a multiple of 32 and the 8 output channels in columns 0..7 of 16 (rest zero);
    lane l holds column l & 15, rows 8 * (l >> 4) .. + 7 of its block.  Returned as int16 bits."""
    Co, Ci = w.shape[0], w.shape[1]
    K = 9 * Ci
    nkb = (K + 31) // 32
    ntn = (max(Co, 16) + 15) // 16
    b = torch.zeros((3, nkb * 32, ntn * 16), dtype=torch.float32, device=w.device)
    # w[n, ci, kz, ky, kx] -> b[kz, (ky*3+kx)*Ci + ci, n]
    b[:, :K, :Co] = w.permute(2, 3, 4, 1, 0).reshape(3, K, Co)
    b = b.reshape(3, nkb, 4, 8, ntn, 16).permute(0, 1, 4, 2, 5, 3)      # [kz][kb][ntile][kgroup][n][j]
    return b.reshape(3, nkb, ntn, 64, 8).to(dt or h16_dtype()).view(torch.int16).contiguous()


def _pack_c8_kzfold_bf16x3(w):
    """[1,Ci,3,3,3] -> the B operands of d3d_conv3d_k3_c1_bf16x3: [hi | mid | lo] x _pack_c8_kzfold_bf16."""
    return torch.stack([_pack_c8_kzfold_bf16(part, torch.bfloat16) for part in _split3_bf16(w)]).contiguous()


def _pack_t2_bf16x3(w):
    """nn.ConvTranspose3d weight [Ci,Co,3,3,3] -> the B operands of d3d_convtranspose3d_k3s2_zs_bf16x3: [hi | mid | lo] x _pack_t2_bf16."""
    return torch.stack([_pack_t2_bf16(part, torch.bfloat16) for part in _split3_bf16(w)]).contiguous()


def _pack_c8_bf16x3(w):
    """[Co,Ci,3,3,3] -> the B operands of d3d_conv3d_k3_zs_bf16x3: [hi | mid | lo] x _pack_c8_bf16 (the exact three-way bf16 split)."""
    return torch.stack([_pack_c8_bf16(part, torch.bfloat16) for part in _split3_bf16(w)]).contiguous()


def _pack_c8_kzfold_bf16(w, dt=None):
    """nn.Conv3d weight [1,Ci,3,3,3] -> B operands of d3d_conv3d_k3_c1_cl_h16: ONE tile per K block whose columns 0, 1, 2 are
    the k_z = 0, 1, 2 slices (K = (k_y, k_x, c_in), padded to a multiple of 32); [K block][lane][8], lane l = column l & 15,
    K rows 8 * (l >> 4) .. + 7.  int16 bits (bf16)."""
    Ci = w.shape[1]
    K = 9 * Ci
    nkb = (K + 31) // 32
    b = torch.zeros((nkb * 32, 16), dtype=torch.float32, device=w.device)
    b[:K, :3] = w[0].permute(2, 3, 0, 1).reshape(K, 3)          # [ci, kz, ky, kx] -> [ky, kx, ci, kz] -> rows (ky*3+kx)*Ci + ci, column kz
    b = b.reshape(nkb, 4, 8, 16).permute(0, 1, 3, 2)            # [kb][kgroup][n][j]
    return b.reshape(nkb, 64, 8).to(dt or h16_dtype()).view(torch.int16).contiguous()


def conv3d_k3(x, weight, scale=None, shift=None, skip=None, relu=True, stride=1):
    """x [Ci,D,H,W], weight [Co,Ci,3,3,3] -> [Co,Do,Ho,Wo] with folded-BN affine, ReLU, skip (after ReLU)."""
    Ci, D, H, W = x.shape
    Co = weight.shape[0]
    if tuple(weight.shape) != (Co, Ci, 3, 3, 3):
        raise ValueError("weight must be [Co,%d,3,3,3] (got %s)" % (Ci, tuple(weight.shape)))
    # C_out = 1 (the probability layer, cas_mvsnet.py:110) has its own streaming VALU kernel behind d3d_conv3d_k3: a
    # single output channel fills 1/16 of a matrix-core tile (D3D_CONV_CO1=0 sends it through the folded MFMA form)
    if stride == 1 and ((Ci in (8, 16, 32) and Co in (8, 16)) or (Ci, Co) in ((32, 32), (64, 64))) and W % 4 == 0 and _use_mfma() \
            and conv_precision() != "h16" and _cfg.get("D3D_CONV_C8X3") != "0":
        # fp32 mode of conv0 and conv2 (cas_mvsnet.py:84,87): the z-streaming matrix-core kernel on three-way bf16 splits of both
        # operands -- fp32 accuracy (six products per K block), each plane staged once.  Against the kernels it replaces
        # (tools/x3_bench.py): conv0 32 -> 8 / 16 -> 8 / 8 -> 8 at the three stage volumes 2.65 / 3.22 / 1.85 -> 2.47 / 2.84 / 1.69 ms,
        # conv2 16 -> 16 0.49 / 0.99 / 0.83 -> 0.18 / 0.43 / 0.44 ms
        wp = derived_weight(weight, "c8bf16x3", _pack_c8_bf16x3)
        out = torch.empty((Co, D, H, W), dtype=torch.float32, device=x.device)
        if skip is not None and skip.shape != out.shape:
            raise ValueError("skip shape %s != output shape %s" % (tuple(skip.shape), tuple(out.shape)))
        rc = _lib.load().d3d_conv3d_k3_zs_bf16x3(_chk(x, "x", 4), ctypes.c_void_p(wp.data_ptr()), _opt(scale, "scale"),
                                                 _opt(shift, "shift"), _opt(skip, "skip"), int(relu), Ci, Co, D, H, W,
                                                 _chk(out, "out"), _stream())
        if rc != _lib.ERR_UNSUPPORTED:
            _lib.check(rc, "d3d_conv3d_k3_zs_bf16x3")
            return out
    if Co == 8 and stride == 1 and Ci % 8 == 0 and _use_mfma() and not _cfg.off("co8") \
            and conv_precision() != "h16" and 7 * D * H * W * 4 + H * W * 4 < 2 ** 31:
        # C_out = 8 (conv0 of every CostRegNet): z-streaming kernel on the fp32 vector units (same peak as the fp32 matrix
        # cores, which an 8-row GEMM half fills); weights re-laid out [Ci][ky][kx][kz][8] once per parameter version
        wp = derived_weight(weight, "co8", lambda w: w.permute(1, 3, 4, 2, 0))
        out = torch.empty((8, D, H, W), dtype=torch.float32, device=x.device)
        if skip is not None and skip.shape != out.shape:
            raise ValueError("skip shape %s != output shape %s" % (tuple(skip.shape), tuple(out.shape)))
        rc = _lib.load().d3d_conv3d_k3_co8(_chk(x, "x", 4), _chk(wp, "wpacked"), _opt(scale, "scale"), _opt(shift, "shift"),
                                           _opt(skip, "skip"), int(relu), Ci, D, H, W, _chk(out, "out"), _stream())
        _lib.check(rc, "d3d_conv3d_k3_co8")
        return out
    if stride == 1 and Ci in (8, 16, 32) and (Co in (8, 16) or (Co == 32 and Ci == 32)) and W % 4 == 0 and _use_mfma() \
            and conv_precision() == "h16" and not _cfg.off("c8"):
        # conv0 / conv2 / conv4 of every CostRegNet with bf16 operands: z-streaming matrix-core kernel (each plane read once)
        wp = derived_weight(weight, "c8bf16", _pack_c8_bf16)
        out = torch.empty((Co, D, H, W), dtype=torch.float32, device=x.device)
        if skip is not None and skip.shape != out.shape:
            raise ValueError("skip shape %s != output shape %s" % (tuple(skip.shape), tuple(out.shape)))
        rc = _lib.load().d3d_conv3d_k3_zs_h16(_chk(x, "x", 4), ctypes.c_void_p(wp.data_ptr()), _opt(scale, "scale"),
                                               _opt(shift, "shift"), _opt(skip, "skip"), int(relu), Ci, Co, D, H, W,
                                               _chk(out, "out"), _stream())
        if rc != _lib.ERR_UNSUPPORTED:
            _lib.check(rc, "d3d_conv3d_k3_zs_h16")
            return out
    if Co == 1 and stride == 1 and Ci == 8 and W % 4 == 0 and _use_mfma() and conv_precision() != "h16" \
            and _cfg.get("D3D_CONV_C8X3") == "all":
        # fp32 mode of the probability layer (cas_mvsnet.py:110) on the k_z-folded matrix-core kernel with split operands: built
        # and tested, but SLOWER than the vector-unit kernel it would replace (0.25 / 0.59 / 0.61 -> 0.34 / 0.87 / 0.85 ms at the
        # three stage volumes: one output channel fills 3 of 16 columns), so only D3D_CONV_C8X3=all routes here
        wf = derived_weight(weight, "c8kzfoldx3", _pack_c8_kzfold_bf16x3)
        out = torch.empty((1, D, H, W), dtype=torch.float32, device=x.device)
        if skip is not None and skip.shape != out.shape:
            raise ValueError("skip shape %s != output shape %s" % (tuple(skip.shape), tuple(out.shape)))
        rc = _lib.load().d3d_conv3d_k3_c1_bf16x3(_chk(x, "x", 4), ctypes.c_void_p(wf.data_ptr()), _opt(scale, "scale"),
                                                 _opt(shift, "shift"), _opt(skip, "skip"), int(relu), Ci, D, H, W,
                                                 _chk(out, "out"), _stream())
        if rc != _lib.ERR_UNSUPPORTED:
            _lib.check(rc, "d3d_conv3d_k3_c1_bf16x3")
            return out
    if stride == 2 and (Ci, Co) in ((8, 16), (16, 32), (32, 64)) and ((W - 1) // 2 + 1) % 4 == 0 and _use_mfma() \
            and conv_precision() != "h16" and _cfg.get("D3D_CONV_C8X3") != "0":
        # fp32 mode of conv1 / conv3 / conv5 (cas_mvsnet.py:86,89,92): the stride-2 z-streaming kernel on three-way bf16 splits
        # (csrc/conv_s2x3.hip) instead of the vector-unit stream kernels
        wp = derived_weight(weight, "c8bf16x3", _pack_c8_bf16x3)
        o = lambda n: (n - 1) // 2 + 1
        out = torch.empty((Co, o(D), o(H), o(W)), dtype=torch.float32, device=x.device)
        if skip is not None and skip.shape != out.shape:
            raise ValueError("skip shape %s != output shape %s" % (tuple(skip.shape), tuple(out.shape)))
        rc = _lib.load().d3d_conv3d_k3s2_zs_bf16x3(_chk(x, "x", 4), ctypes.c_void_p(wp.data_ptr()), _opt(scale, "scale"),
                                                   _opt(shift, "shift"), _opt(skip, "skip"), int(relu), Ci, Co, D, H, W,
                                                   _chk(out, "out"), _stream())
        if rc != _lib.ERR_UNSUPPORTED:
            _lib.check(rc, "d3d_conv3d_k3s2_zs_bf16x3")
            dispatch_counts["conv3d_s2_x3"] += 1
            return out
    co1 = Co == 1 and stride == 1 and Ci == 8 and not _cfg.off("co1")
    if _use_mfma() and Co <= 64 and not co1:
        y = conv_k3_mfma(x, weight, scale, shift, skip, act=1 if relu else 0, stride=stride)
        if y is not None:
            return y
    o = lambda n: (n - 1) // stride + 1
    out = torch.empty((Co, o(D), o(H), o(W)), dtype=torch.float32, device=x.device)
    if skip is not None and skip.shape != out.shape:
        raise ValueError("skip shape %s != output shape %s" % (tuple(skip.shape), tuple(out.shape)))
    rc = _lib.load().d3d_conv3d_k3(_chk(x, "x", 4), _chk(weight, "weight"), _opt(scale, "scale"),
                                   _opt(shift, "shift"), _opt(skip, "skip"), int(relu), Ci, Co, D, H, W, stride,
                                   _chk(out, "out"), _stream())
    _lib.check(rc, "d3d_conv3d_k3")
    return out


def _pack_t2_bf16(w, dt=None):
    """nn.ConvTranspose3d weight [Ci,Co,3,3,3] -> B operands of v_mfma_f32_16x16x32_bf16 for d3d_convtranspose3d_k3s2_zs_h16.
    Output parity class (pz,py,px), in the order pz*4 + py*2 + px: taps (dz,dy,dx), d <= p per dimension, enumerated dz-major;
    an even output coordinate uses kernel index 1 (d = 0), an odd one index 2 (d = 0) and 0 (d = 1).  K = (tap, ci) padded to
    a multiple of 32, output channels padded to a multiple of 16; per class [K block][N tile][lane][8] with lane l holding
    column l & 15 and rows 8 * (l >> 4) .. + 7 of its block.  Returned as int16 bits (bf16)."""
    Ci, Co = w.shape[0], w.shape[1]
    ntn = (max(Co, 16) + 15) // 16
    kmap = {(0, 0): 1, (1, 0): 2, (1, 1): 0}
    parts = []
    for p in range(8):
        pz, py, px = p >> 2, (p >> 1) & 1, p & 1
        taps = [(dz, dy, dx) for dz in range(1 + pz) for dy in range(1 + py) for dx in range(1 + px)]
        K = len(taps) * Ci
        nkb = (K + 31) // 32
        b = torch.zeros((nkb * 32, ntn * 16), dtype=torch.float32, device=w.device)
        for t, (dz, dy, dx) in enumerate(taps):
            b[t * Ci:(t + 1) * Ci, :Co] = w[:, :, kmap[(pz, dz)], kmap[(py, dy)], kmap[(px, dx)]]
        # [kb][kgroup][j][nt][n] -> [kb][nt][kgroup][n][j]
        b = b.reshape(nkb, 4, 8, ntn, 16).permute(0, 3, 1, 4, 2)
        parts.append(b.reshape(nkb * ntn * 64, 8))
    return torch.cat(parts).to(dt or h16_dtype()).view(torch.int16).contiguous()


def _pack_t2_fold_bf16(w, dt=None):
    """ConvTranspose3d weight [Ci,8,3,3,3] -> A operands of the x-folded form of d3d_convtranspose3d_k3s2_cl_h16: per output
    parity class (pz,py), K = (taps (dz,dy,dx) with dx in {0,1}, dz-major) x ci, GEMM row r = px * 8 + channel: the even
    column (px = 0) uses kernel column 1 of the dx = 0 taps (its dx = 1 entries are zero), the odd one column 2 (dx = 0)
    and 0 (dx = 1).  [class][K block][lane][8], lane l = row l & 15, K rows 8 * (l >> 4) .. + 7.  int16 bits (bf16)."""
    Ci, Co = w.shape[0], w.shape[1]
    assert Co == 8
    kmap = {(0, 0): 1, (1, 0): 2, (1, 1): 0}
    parts = []
    for c in range(4):
        pz, py = c >> 1, c & 1
        taps = [(dz, dy, dx) for dz in range(1 + pz) for dy in range(1 + py) for dx in range(2)]
        K = len(taps) * Ci
        nkb = (K + 31) // 32
        b = torch.zeros((nkb * 32, 16), dtype=torch.float32, device=w.device)
        for t, (dz, dy, dx) in enumerate(taps):
            kz, ky = kmap[(pz, dz)], kmap[(py, dy)]
            if dx == 0:
                b[t * Ci:(t + 1) * Ci, 0:8] = w[:, :, kz, ky, 1]
            b[t * Ci:(t + 1) * Ci, 8:16] = w[:, :, kz, ky, kmap[(1, dx)]]
        b = b.reshape(nkb, 4, 8, 16).permute(0, 1, 3, 2)      # [kb][kgroup][row][j]
        parts.append(b.reshape(nkb * 64, 8))
    return torch.cat(parts).to(dt or h16_dtype()).view(torch.int16).contiguous()


def convtranspose3d_k3s2(x, weight, scale=None, shift=None, skip=None, relu=True):
    """x [Ci,D,H,W], weight [Ci,Co,3,3,3] -> [Co,2D,2H,2W]."""
    Ci, D, H, W = x.shape
    Co = weight.shape[1]
    if tuple(weight.shape) != (Ci, Co, 3, 3, 3):
        raise ValueError("weight must be [%d,Co,3,3,3] (got %s)" % (Ci, tuple(weight.shape)))
    if (Ci, Co) in ((16, 8), (16, 16), (32, 16), (64, 32)) and _use_mfma() and conv_precision() != "h16" and _cfg.get("D3D_CONV_C8X3") != "0":
        # fp32 mode of conv11, conv9 and conv7 (cas_mvsnet.py:103, 100, 97: 16 -> 8 to the full-resolution volume, 32 -> 16, 64 -> 32): the per-parity matrix-core kernel on
        # three-way bf16 splits of both operands (fp32 accuracy, see conv3d_k3)
        wp = derived_weight(weight, "t2bf16x3", _pack_t2_bf16x3)
        out = torch.empty((Co, 2 * D, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
        if skip is not None and skip.shape != out.shape:
            raise ValueError("skip shape %s != output shape %s" % (tuple(skip.shape), tuple(out.shape)))
        rc = _lib.load().d3d_convtranspose3d_k3s2_zs_bf16x3(_chk(x, "x", 4), ctypes.c_void_p(wp.data_ptr()), _opt(scale, "scale"),
                                                            _opt(shift, "shift"), _opt(skip, "skip"), int(relu), Ci, Co, D, H, W,
                                                            _chk(out, "out"), _stream())
        if rc != _lib.ERR_UNSUPPORTED:
            _lib.check(rc, "d3d_convtranspose3d_k3s2_zs_bf16x3")
            return out
    if Co == 8 and Ci % 8 == 0 and _use_mfma() and not _cfg.off("co8") \
            and conv_precision() != "h16" and 7 * D * H * W * 4 + H * W * 4 < 2 ** 31:
        # C_out = 8 (conv11 of every CostRegNet): z-streaming kernel on the fp32 vector units, weights [Ci][kz][ky][kx][8]
        wp = derived_weight(weight, "coT8", lambda w: w.permute(0, 2, 3, 4, 1))
        out = torch.empty((8, 2 * D, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
        if skip is not None and skip.shape != out.shape:
            raise ValueError("skip shape %s != output shape %s" % (tuple(skip.shape), tuple(out.shape)))
        rc = _lib.load().d3d_convtranspose3d_k3s2_co8(_chk(x, "x", 4), _chk(wp, "wpacked"), _opt(scale, "scale"),
                                                      _opt(shift, "shift"), _opt(skip, "skip"), int(relu), Ci, D, H, W,
                                                      _chk(out, "out"), _stream())
        _lib.check(rc, "d3d_convtranspose3d_k3s2_co8")
        return out
    if (Ci, Co) in ((16, 8), (16, 16), (32, 16), (64, 32)) and _use_mfma() and conv_precision() == "h16" \
            and not _cfg.off("t2"):
        # decoder layers of CostRegNet with bf16 operands: eight per-parity dense convolutions on the matrix cores, z-streaming
        wp = derived_weight(weight, "t2bf16", _pack_t2_bf16)
        out = torch.empty((Co, 2 * D, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
        if skip is not None and skip.shape != out.shape:
            raise ValueError("skip shape %s != output shape %s" % (tuple(skip.shape), tuple(out.shape)))
        rc = _lib.load().d3d_convtranspose3d_k3s2_zs_h16(_chk(x, "x", 4), ctypes.c_void_p(wp.data_ptr()), _opt(scale, "scale"),
                                                          _opt(shift, "shift"), _opt(skip, "skip"), int(relu), Ci, Co, D, H, W,
                                                          _chk(out, "out"), _stream())
        if rc != _lib.ERR_UNSUPPORTED:
            _lib.check(rc, "d3d_convtranspose3d_k3s2_zs_h16")
            return out
    if _use_mfma() and Co <= 64:
        y = convtranspose_k3s2_mfma(x, weight, scale, shift, skip, act=1 if relu else 0)
        if y is not None:
            return y
    out = torch.empty((Co, 2 * D, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    if skip is not None and skip.shape != out.shape:
        raise ValueError("skip shape %s != output shape %s" % (tuple(skip.shape), tuple(out.shape)))
    rc = _lib.load().d3d_convtranspose3d_k3s2(_chk(x, "x", 4), _chk(weight, "weight"), _opt(scale, "scale"),
                                              _opt(shift, "shift"), _opt(skip, "skip"), int(relu), Ci, Co, D, H, W,
                                              _chk(out, "out"), _stream())
    _lib.check(rc, "d3d_convtranspose3d_k3s2")
    return out


# ----------------------------------------------------------------------------------------
# Channel-last bf16 volumes: what the CostRegNet layers hand to each other in bf16 mode (BASELINE config 3).
# A "CL" volume is a 16-bit tensor (h16_dtype(): torch.float16 unless the library was built for bfloat16) [D,H,W,C]; a planar one the usual float32 [C,D,H,W].
# ----------------------------------------------------------------------------------------
def channel_last_enabled():
    return not _cfg.off("cl")


def _chk_cl(t, name, cl8=False):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == h16_dtype() and t.is_contiguous()
            and (t.dim() == 4 or (cl8 and t.dim() == 5 and t.shape[4] == 8))):
        raise TypeError("%s must be a contiguous CUDA 16-bit (ops.h16_dtype()) tensor [D,H,W,C]%s" % (name, " or [D,C/8,H,W,8]" if cl8 else ""))
    return ctypes.c_void_p(t.data_ptr())


def to_cl(x):
    """planar fp32 [C,D,H,W] -> channel-last bf16 [D,H,W,C] (round to nearest even)."""
    C, D, H, W = x.shape
    if C % 8:
        raise ValueError("channel-last volumes need C % 8 == 0 (got %d)" % C)
    out = torch.empty((D, H, W, C), dtype=h16_dtype(), device=x.device)
    _lib.check(_lib.load().d3d_volume_planar_to_cl_h16(_chk(x, "x", 4), C, D * H * W, _chk_cl(out, "out"), _stream()),
               "d3d_volume_planar_to_cl_h16")
    return out


def from_cl(x):
    """channel-last bf16 [D,H,W,C] -> planar fp32 [C,D,H,W] (exact)."""
    D, H, W, C = x.shape
    if C % 8:
        raise ValueError("channel-last volumes need C % 8 == 0 (got %d)" % C)
    out = torch.empty((C, D, H, W), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().d3d_volume_cl_h16_to_planar(_chk_cl(x, "x"), C, D * H * W, _chk(out, "out"), _stream()),
               "d3d_volume_cl_h16_to_planar")
    return out


def _planar(t):
    return None if t is None else (from_cl(t) if t.dtype == h16_dtype() else t)


def conv3d_k3_cl(x, weight, scale=None, shift=None, skip=None, relu=True, stride=1, out_cl=True):
    """conv3d_k3 with bf16 operands on either activation format: x planar fp32 [Ci,D,H,W] or CL bf16 [D,H,W,Ci];
    returns CL [Do,Ho,Wo,Co] (out_cl) or planar fp32 [Co,Do,Ho,Wo]; `skip` comes in the output's format.  Shapes the
    channel-last kernels do not take go through the planar kernels and the two format conversions."""
    in_cl = x.dtype == h16_dtype()
    cl8 = in_cl and x.dim() == 5   # [D,Ci/8,H,W,8]: the sweep kernels' CL8 volume (stride-1 layers on the conv0 kernel family)
    if cl8 and (stride != 1 or x.shape[4] != 8):
        raise ValueError("a CL8 input [D,Ci/8,H,W,8] is taken by the stride-1 layers only")
    (D, H, W, Ci) = ((x.shape[0], x.shape[2], x.shape[3], x.shape[1] * 8) if cl8 else x.shape) if in_cl \
        else (x.shape[1], x.shape[2], x.shape[3], x.shape[0])
    Co = weight.shape[0]
    if tuple(weight.shape) != (Co, Ci, 3, 3, 3):
        raise ValueError("weight must be [Co,%d,3,3,3] (got %s)" % (Ci, tuple(weight.shape)))
    o = lambda n: (n - 1) // stride + 1
    oshape = (o(D), o(H), o(W), Co) if out_cl else (Co, o(D), o(H), o(W))
    if skip is not None and (tuple(skip.shape) != oshape or (skip.dtype == h16_dtype()) != bool(out_cl)):
        raise ValueError("skip %s %s does not match the output %s" % (skip.dtype, tuple(skip.shape), oshape))
    xp = _chk_cl(x, "x", cl8=True) if in_cl else _chk(x, "x", 4)
    fmt = 2 if cl8 else int(in_cl)
    sp = None if skip is None else (_chk_cl(skip, "skip") if out_cl else _chk(skip, "skip"))
    wp = derived_weight(weight, "c8bf16", _pack_c8_bf16)
    wptr = ctypes.c_void_p(wp.data_ptr())
    out = torch.empty(oshape, dtype=h16_dtype() if out_cl else torch.float32, device=x.device)
    optr = ctypes.c_void_p(out.data_ptr())
    rc = _lib.ERR_UNSUPPORTED
    if stride == 1 and Co == 1 and not out_cl and not _cfg.off("kzfold"):
        # the probability layer: k_z folded into the columns of one operand tile
        wf = derived_weight(weight, "c8kzfold", _pack_c8_kzfold_bf16)
        rc = _lib.load().d3d_conv3d_k3_c1_cl_h16(xp, fmt, ctypes.c_void_p(wf.data_ptr()), _opt(scale, "scale"),
                                                  _opt(shift, "shift"), sp, int(relu), Ci, D, H, W, optr, _stream())
        if rc != _lib.ERR_UNSUPPORTED:
            _lib.check(rc, "d3d_conv3d_k3_c1_cl_h16")
            dispatch_counts["conv3d_cl"] += 1
            return out
    if stride == 1:
        rc = _lib.load().d3d_conv3d_k3_cl_h16(xp, fmt, wptr, _opt(scale, "scale"), _opt(shift, "shift"), sp, int(relu),
                                               Ci, Co, D, H, W, optr, int(out_cl), _stream())
    elif stride == 2 and in_cl and out_cl:
        rc = _lib.load().d3d_conv3d_k3s2_cl_h16(xp, wptr, _opt(scale, "scale"), _opt(shift, "shift"), sp, int(relu), Ci, Co,
                                                 D, H, W, optr, _stream())
    if rc != _lib.ERR_UNSUPPORTED:
        _lib.check(rc, "d3d_conv3d_k3_cl_h16" if stride == 1 else "d3d_conv3d_k3s2_cl_h16")
        dispatch_counts["conv3d_cl8_in" if cl8 else "conv3d_cl"] += 1
        return out
    dispatch_counts["conv3d_cl_fallback"] += 1
    saved = _cfg.state.conv_precision
    _cfg.state.conv_precision = "h16"
    try:
        y = conv3d_k3(_planar(cl8_to_cl(x) if cl8 else x), weight, scale, shift, _planar(skip), relu=relu, stride=stride)
    finally:
        _cfg.state.conv_precision = saved
    return to_cl(y) if out_cl else y


def convtranspose3d_k3s2_cl(x, weight, scale=None, shift=None, skip=None, relu=True):
    """convtranspose3d_k3s2 with bf16 operands on channel-last volumes: x [D,H,W,Ci] bf16, skip / result [2D,2H,2W,Co] bf16."""
    D, H, W, Ci = x.shape
    Co = weight.shape[1]
    if tuple(weight.shape) != (Ci, Co, 3, 3, 3):
        raise ValueError("weight must be [%d,Co,3,3,3] (got %s)" % (Ci, tuple(weight.shape)))
    oshape = (2 * D, 2 * H, 2 * W, Co)
    if skip is not None and (tuple(skip.shape) != oshape or skip.dtype != h16_dtype()):
        raise ValueError("skip %s %s does not match the output %s" % (skip.dtype, tuple(skip.shape), oshape))
    if (Ci, Co) in ((16, 8), (16, 16), (32, 16), (64, 32)):
        fold = (Ci, Co) == (16, 8) and not _cfg.off("t2fold")   # conv11: both column parities in one GEMM
        wp = derived_weight(weight, "t2foldbf16", _pack_t2_fold_bf16) if fold else derived_weight(weight, "t2bf16", _pack_t2_bf16)
        out = torch.empty(oshape, dtype=h16_dtype(), device=x.device)
        rc = _lib.load().d3d_convtranspose3d_k3s2_cl_h16(_chk_cl(x, "x"), ctypes.c_void_p(wp.data_ptr()), _opt(scale, "scale"),
                                                          _opt(shift, "shift"), None if skip is None else _chk_cl(skip, "skip"),
                                                          int(relu), Ci, Co, D, H, W, ctypes.c_void_p(out.data_ptr()),
                                                          2 if fold else 1, _stream())
        if rc != _lib.ERR_UNSUPPORTED:
            _lib.check(rc, "d3d_convtranspose3d_k3s2_cl_h16")
            dispatch_counts["convtranspose3d_cl"] += 1
            return out
    dispatch_counts["conv3d_cl_fallback"] += 1
    saved = _cfg.state.conv_precision
    _cfg.state.conv_precision = "h16"
    try:
        y = convtranspose3d_k3s2(from_cl(x), weight, scale, shift, _planar(skip), relu=relu)
    finally:
        _cfg.state.conv_precision = saved
    return to_cl(y)


def convtranspose3d_prob_cl(x, weight, scale, shift, skip, prob_weight, prob_bias, relu=True):
    """conv11 + prob of a CostRegNet (cas_mvsnet.py:103-105,118-119) in one launch: x CL [D,H,W,16] bf16, skip CL
    [2D,2H,2W,8] bf16 -> planar fp32 [2D,2H,2W] = Conv3d_8->1(skip + act(scale * ConvTranspose3d_16->8(x) + shift)) + bias,
    the 8-channel volume rounded to bf16 as between the two unfused layers (bit-identical to them) but kept in LDS
    (d3d_convtranspose3d_prob_cl_h16).  None for shapes the kernel does not take."""
    D, H, W, Ci = x.shape
    if _cfg.off("t2prob") or _cfg.off("t2fold") or _cfg.off("kzfold") or Ci != 16 or W % 2 \
            or tuple(weight.shape) != (16, 8, 3, 3, 3) or tuple(prob_weight.shape) != (1, 8, 3, 3, 3):
        return None
    oshape = (2 * D, 2 * H, 2 * W)
    if skip is not None and (tuple(skip.shape) != oshape + (8,) or skip.dtype != h16_dtype()):
        raise ValueError("skip %s %s does not match the output %s" % (skip.dtype, tuple(skip.shape), oshape + (8,)))
    wt = derived_weight(weight, "t2foldbf16", _pack_t2_fold_bf16)
    wp = derived_weight(prob_weight, "c8kzfold", _pack_c8_kzfold_bf16)
    out = torch.empty(oshape, dtype=torch.float32, device=x.device)
    rc = _lib.load().d3d_convtranspose3d_prob_cl_h16(
        _chk_cl(x, "x"), ctypes.c_void_p(wt.data_ptr()), _opt(scale, "scale"), _opt(shift, "shift"),
        None if skip is None else _chk_cl(skip, "skip"), int(relu), ctypes.c_void_p(wp.data_ptr()), _opt(prob_bias, "prob_bias"),
        D, H, W, ctypes.c_void_p(out.data_ptr()), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "d3d_convtranspose3d_prob_cl_h16")
    dispatch_counts["convtranspose3d_prob_cl"] += 1
    return out


def conv1x1_upskip(x, weight, bias, coarse):
    """conv1x1(x) + bias + nearest-x2 upsampling of `coarse` (FPN lateral, module.py:736-747) in one pass.
    x [Ci,H,W], weight [Co,Ci,1,1], coarse [Co,H/2,W/2].  Returns None for shapes the kernel does not take."""
    Ci, H, W = x.shape
    Co = weight.shape[0]
    if (Ci, Co) not in ((8, 32), (16, 32)) or H % 2 or W % 2 or tuple(coarse.shape) != (Co, H // 2, W // 2) \
            or tuple(weight.shape) != (Co, Ci, 1, 1) or _cfg.off("upskip"):
        return None
    wp = derived_weight(weight, "c11", lambda w: w.reshape(Co, Ci).t())
    out = torch.empty((Co, H, W), dtype=torch.float32, device=x.device)
    rc = _lib.load().d3d_conv1x1_upskip(_chk(x, "x", 3), Ci, _chk(wp, "wpacked"), _opt(bias, "bias"), _chk(coarse, "coarse", 3),
                                        Co, H, W, _chk(out, "out"), _stream())
    _lib.check(rc, "d3d_conv1x1_upskip")
    return out


_CONV2D_STREAM_MIN = 256 * 256


def conv2d_stream(x, weight, scale, shift, skip, act, x2=None, aux1=None, ep_split=0):
    """3x3 stride-1 convolution over cat(x, x2) with 8 | 16 output channels on the vector-unit streaming kernel
    (d3d_conv2d_k3_stream): large feature-pyramid layers (act 0 | 1) and conv-GRU cells (act 2 | 3, see
    gru_cell_fused).  Returns None when the layer is not one of those."""
    Ci0, H, W = x.shape
    Ci1 = 0 if x2 is None else x2.shape[0]
    Co = weight.shape[0]
    if (Co not in (8, 16) or not _use_mfma() or conv_precision() == "h16" or H * W < _CONV2D_STREAM_MIN
            or 8 * H * W * 4 >= 2 ** 31 or _cfg.off("conv2d_stream")
            or (x2 is not None and Ci0 % 8 != 0) or tuple(weight.shape) != (Co, Ci0 + Ci1, 3, 3)):
        return None

    def pack(w):
        ci = Ci0 + Ci1
        wp = w.new_zeros(((ci + 7) // 8 * 8, 3, 3, Co))
        wp[:ci] = w.permute(1, 2, 3, 0)
        return wp
    wp = derived_weight(weight, "c2s", pack)
    out = torch.empty((Co, H, W), dtype=torch.float32, device=x.device)
    if act <= 1 and skip is not None and skip.shape != out.shape:
        raise ValueError("skip shape mismatch")
    rc = _lib.load().d3d_conv2d_k3_stream(_chk(x, "x", 3), Ci0, _opt(x2, "x2"), Ci1, _chk(wp, "wpacked"),
                                          _opt(scale, "scale"), _opt(shift, "shift"), _opt(skip, "skip"), _opt(aux1, "aux1"),
                                          int(ep_split), int(act), Co, H, W, _chk(out, "out"), _stream())
    _lib.check(rc, "d3d_conv2d_k3_stream")
    return out  # pixels from which the vector-unit streaming form of conv2d_k3 is used


def _pack_z2_bf16(w, dt=None):
    """nn.Conv2d weight [Co,Ci,3,3] -> B operands of v_mfma_f32_16x16x32_bf16 for d3d_conv2d_k3_zs_h16: K = (k_y, k_x, c_in)
    padded to a multiple of 32, output channels to a multiple of 16; [K block][N tile][lane][8], lane l = column l & 15,
    K rows 8 * (l >> 4) .. + 7 of its block.  int16 bits (bf16)."""
    Co, Ci = w.shape[0], w.shape[1]
    K = w.shape[2] * w.shape[3] * Ci                                   # (3 x 3; 5 x 5 for d3d_conv2d_k5s2_zs_bf16x3)
    nkb = (K + 31) // 32
    ntn = (max(Co, 16) + 15) // 16
    b = torch.zeros((nkb * 32, ntn * 16), dtype=torch.float32, device=w.device)
    b[:K, :Co] = w.permute(2, 3, 1, 0).reshape(K, Co)                  # [ky, kx, ci, co]
    b = b.reshape(nkb, 4, 8, ntn, 16).permute(0, 3, 1, 4, 2)           # [kb][ntile][kgroup][n][j]
    return b.reshape(nkb, ntn, 64, 8).to(dt or h16_dtype()).view(torch.int16).contiguous()


def _pack_z2_f32(w):
    """nn.Conv2d weight [Co,Ci,3,3] -> fp32 B operands of v_mfma_f32_16x16x4_f32 for d3d_conv2d_k3_zs_f32: K = (k_y, k_x, c_in)
    in blocks of 4, output channels padded to a multiple of 16; [K block][N tile][lane], lane l = column l & 15, K row l >> 4."""
    Co, Ci = w.shape[0], w.shape[1]
    K = 9 * Ci
    ntn = (max(Co, 16) + 15) // 16
    b = torch.zeros((K, ntn * 16), dtype=torch.float32, device=w.device)
    b[:, :Co] = w.permute(2, 3, 1, 0).reshape(K, Co)
    return b.reshape(K // 4, 4, ntn, 16).permute(0, 2, 1, 3).reshape(K // 4, ntn, 64).contiguous()


def avgpool_4_8(x):
    """AvgPool2d(4) and AvgPool2d(8) of x [C,H,W] in one read (d3d_avgpool2d_4_8) -> ([C,H//4,W//4], [C,H//8,W//8]); None for
    shapes the kernel does not take."""
    C, H, W = x.shape
    if H < 8 or W < 8 or W % 4 or _cfg.off("context_fused"):
        return None
    o4 = torch.empty((C, H // 4, W // 4), dtype=torch.float32, device=x.device)
    o8 = torch.empty((C, H // 8, W // 8), dtype=torch.float32, device=x.device)
    rc = _lib.load().d3d_avgpool2d_4_8(_chk(x, "x", 3), C, H, W, _chk(o4, "out4"), _chk(o8, "out8"), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "d3d_avgpool2d_4_8")
    return o4, o8


def conv1x1_context(f, weight, a, b):
    """out = weight [Co,Ci] . f [Ci,H,W] + bilinear_resize(a [Co,Ha,Wa]) + bilinear_resize(b [Co,Hb,Wb]) (align_corners=False)
    in one pass over f (d3d_conv1x1_context: the pooled-context heads of the AdaMVS pyramid); None for other shapes."""
    Ci, H, W = f.shape
    Co = weight.shape[0]
    if (tuple(weight.shape) != (Co, Ci) or Ci != Co or Ci not in (8, 16, 32) or W % 4 or a.shape[0] != Co or b.shape[0] != Co
            or 3 * a.shape[2] > W or 3 * b.shape[2] > W or _cfg.off("context_fused")):
        return None
    out = torch.empty((Co, H, W), dtype=torch.float32, device=f.device)
    rc = _lib.load().d3d_conv1x1_context(_chk(f, "f", 3), Ci, _chk(weight, "weight"), _chk(a, "a", 3), a.shape[1], a.shape[2],
                                         _chk(b, "b", 3), b.shape[1], b.shape[2], Co, H, W, _chk(out, "out"), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "d3d_conv1x1_context")
    return out


def conv3x3_bias_border_(out, taps):
    """In place: out [Co,H,W] -= the taps [Co,3,3] whose source pixel lies outside the image, on the one-pixel border
    (d3d_conv3x3_bias_border)."""
    Co, H, W = out.shape
    if tuple(taps.shape) != (Co, 3, 3):
        raise ValueError("taps must be [%d,3,3]" % Co)
    rc = _lib.load().d3d_conv3x3_bias_border(_chk(out, "out", 3), _chk(taps, "taps"), Co, H, W, _stream())
    _lib.check(rc, "d3d_conv3x3_bias_border")
    return out


def _split3_bf16(w):
    """fp32 tensor -> its exact three-way bf16 split (hi, mid, lo as fp32 tensors; hi + mid + lo == w in fp32)."""
    w = w.to(torch.float32)
    hi = w.to(torch.bfloat16).to(torch.float32)
    mid = (w - hi).to(torch.bfloat16).to(torch.float32)
    lo = (w - hi - mid).to(torch.bfloat16).to(torch.float32)
    return hi, mid, lo


def _pack_z2_bf16x3(w):
    """nn.Conv2d weight [Co,Ci,3,3] -> the B operands of d3d_conv2d_k3_zs_bf16x3: [hi | mid | lo] x _pack_z2_bf16."""
    return torch.stack([_pack_z2_bf16(part, torch.bfloat16) for part in _split3_bf16(w)]).contiguous()


def _pack_t2d_bf16x3(w):
    """nn.ConvTranspose2d weight [Ci,Co,3,3] -> the B operands of d3d_convtranspose2d_k3s2_zs_bf16x3: [hi | mid | lo] x _pack_t2d_bf16."""
    return torch.stack([_pack_t2d_bf16(part, torch.bfloat16) for part in _split3_bf16(w)]).contiguous()


def _pack_t2d_k4_bf16(w, dt=None):
    """nn.ConvTranspose2d weight [Ci,Co,4,4] (stride 2, padding 1) -> B operands of the k = 4 transposed tile kernel: per output
    parity class (py,px), order py*2 + px, taps (dy,dx) in {0,1}^2 dy-major; output 2i + p reads input i - 1 + p + d through
    kernel index 3 - p - 2d.  K = (tap, ci), 16 output columns; [K block][lane][8] (bf16 bits)."""
    Ci, Co = w.shape[0], w.shape[1]
    parts = []
    for c in range(4):
        py, px = c >> 1, c & 1
        K = 4 * Ci
        nkb = (K + 31) // 32
        b = torch.zeros((nkb * 32, 16), dtype=torch.float32, device=w.device)
        for t, (dy, dx) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
            b[t * Ci:(t + 1) * Ci, :Co] = w[:, :, 3 - py - 2 * dy, 3 - px - 2 * dx]
        b = b.reshape(nkb, 4, 8, 16).permute(0, 1, 3, 2)                 # [kb][kgroup][n][j]
        parts.append(b.reshape(nkb * 64, 8))
    return torch.cat(parts).to(dt or h16_dtype()).view(torch.int16).contiguous()


def _pack_t2d_k4fold_bf16(w, dt=None):
    """The k = 4 transposed weight [Ci,Co<=8,4,4] with both column parities in one 16-column tile: per row parity py, K =
    (dy in {0,1}, patch column dxx in {0,1,2}, ci); columns 0..7 = even output column (dxx = dx), 8..15 = odd one (dxx = 1 + dx)."""
    Ci, Co = w.shape[0], w.shape[1]
    parts = []
    for py in range(2):
        nkb = (6 * Ci + 31) // 32
        b = torch.zeros((nkb * 32, 16), dtype=torch.float32, device=w.device)
        for dy in range(2):
            for dxx in range(3):
                t = dy * 3 + dxx
                for px in range(2):
                    dx = dxx - px
                    if dx in (0, 1):
                        b[t * Ci:(t + 1) * Ci, px * 8:px * 8 + Co] = w[:, :, 3 - py - 2 * dy, 3 - px - 2 * dx]
        b = b.reshape(nkb, 4, 8, 16).permute(0, 1, 3, 2)
        parts.append(b.reshape(nkb * 64, 8))
    return torch.cat(parts).to(dt or h16_dtype()).view(torch.int16).contiguous()


def _pack_t2d_k4_bf16x3(w):
    """[hi | mid | lo] x the k = 4 packing d3d_convtranspose2d_k4s2_zs_bf16x3 takes: column-folded for C_out <= 8."""
    pack = _pack_t2d_k4fold_bf16 if w.shape[1] <= 8 else _pack_t2d_k4_bf16
    return torch.stack([pack(part, torch.bfloat16) for part in _split3_bf16(w)]).contiguous()


def upsampled_conv_weight(w3):
    """Conv2d weight [Co,Ci,3,3] (padding 1) -> the ConvTranspose2d weight [Ci,Co,4,4] (stride 2, padding 1) with
    conv_transpose2d(f, .) == conv2d(nearest_x2(f), w3): the three taps of an output pixel along an axis fall on two cells
    of f, and the weights of taps sharing a cell add (kernel index 3: tap 0; 1: taps 1 + 2; 0: tap 2; 2: taps 0 + 1)."""
    w = w3.detach().to(torch.float64)
    rows = torch.stack([w[:, :, 2], w[:, :, 1] + w[:, :, 2], w[:, :, 0] + w[:, :, 1], w[:, :, 0]], 2)          # [Co,Ci,4(ky),3]
    full = torch.stack([rows[..., 2], rows[..., 1] + rows[..., 2], rows[..., 0] + rows[..., 1], rows[..., 0]], 3)   # [Co,Ci,4,4]
    return full.permute(1, 0, 2, 3).to(torch.float32).contiguous()


def convtranspose2d_k4_zs(x, weight, scale=None, shift=None, skip=None, act=0, skip_after_act=False):
    """ConvTranspose2d(k 4, stride 2, pad 1): x [Ci,H,W], weight [Ci,Co,4,4] -> [Co,2H,2W] on the transposed tile kernel with
    split operands (fp32 accuracy; d3d_convtranspose2d_k4s2_zs_bf16x3); None for shapes it does not take."""
    Ci, H, W = x.shape
    Co = weight.shape[1]
    if Ci not in (8, 16, 32) or Co > 16 or W % 4 or act not in (0, 1) or tuple(weight.shape) != (Ci, Co, 4, 4) \
            or _cfg.off("conv2d_zs"):
        return None
    wp = derived_weight(weight, "t2dk4x3", _pack_t2d_k4_bf16x3)
    out = torch.empty((Co, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    if skip is not None and skip.shape != out.shape:
        raise ValueError("skip shape mismatch")
    rc = _lib.load().d3d_convtranspose2d_k4s2_zs_bf16x3(
        _chk(x, "x", 3), ctypes.c_void_p(wp.data_ptr()), _opt(scale, "scale"), _opt(shift, "shift"), _opt(skip, "skip"), int(act),
        int(bool(skip_after_act)), Ci, Co, H, W, _chk(out, "out"), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "d3d_convtranspose2d_k4s2_zs_bf16x3")
    return out


def _z2_fp32_entry():
    """fp32-mode flavour of the 2-D tile kernels: "x3" (three-way bf16 split, the default) | "f32" (v_mfma_f32_16x16x4_f32)."""
    return "f32" if _cfg.get("D3D_CONV2D_FP32") == "f32" else "x3"


def conv2d_zs(x, weight, scale=None, shift=None, skip=None, act=0, x2=None, aux1=None, ep_split=0, skip_after_act=False, gn=None):
    """3x3 stride-1 conv over cat(x, x2) on the tile kernel with the fused epilogues of the slice regularisers (act 0 | 1 |
    2 GRU gates | 3 GRU update: see d3d_conv2d_k3_zs_h16) -- bf16 matrix-core operands in bf16 mode; fp32 accuracy otherwise
    (d3d_conv2d_k3_zs_bf16x3: three-way bf16 splits, or with D3D_CONV2D_FP32=f32 the fp32 instruction of d3d_conv2d_k3_zs_f32).  Returns None for shapes the kernel does not take."""
    Ci0, H, W = x.shape
    Ci1 = 0 if x2 is None else x2.shape[0]
    Co = weight.shape[0]
    Ci, bf16 = Ci0 + Ci1, conv_precision() == "h16"
    ok = (Ci in (8, 16, 32) and Co <= 32) or (Ci == 48 and Co <= 48) or (bf16 and Ci in (24, 40) and Co <= 16)   # 48 -> 48: the pair-visibility UNet; 24 | 40: RED-Net's conv_gru1 at stages 2 / 1
    if ok and W % 4 and x2 is None and aux1 is None and act in (0, 1) and H * W <= 256 * 256 and Ci0 % 8 == 0:
        # small images whose width is not a multiple of 4 (the coarsest UNet level, 58 x 86): zero columns on the right are the
        # layer's own padding, so the padded image gives the same outputs; the extra columns are dropped
        pw = (-W) % 4
        y = conv2d_zs(torch.nn.functional.pad(x, (0, pw)), weight, scale, shift,
                      None if skip is None else torch.nn.functional.pad(skip, (0, pw)), act, skip_after_act=skip_after_act)
        return None if y is None else y[:, :, :W].contiguous()
    if not ok or Ci0 % 8 or Ci1 % 8 or W % 4 or _cfg.off("conv2d_zs"):
        return None
    if tuple(weight.shape) != (Co, Ci0 + Ci1, 3, 3):
        raise ValueError("weight must be [Co,%d,3,3] (got %s)" % (Ci0 + Ci1, tuple(weight.shape)))
    lib = _lib.load()
    if bf16 and gn is not None and aux1 is None and Ci in (16, 24, 32, 40) and shift is not None:
        ga = _gn_args(gn, Co, act, scale, skip, x.device)
        if ga is not None:   # the layer and the GroupNorm statistics of its output in one launch
            wp = derived_weight(weight, "z2bf16", _pack_z2_bf16)
            out = torch.empty((Co, H, W), dtype=torch.float32, device=x.device)
            rc = lib.d3d_conv2d_k3_zs_h16_gn(_chk(x, "x", 3), Ci0, _opt(x2, "x2"), Ci1, ctypes.c_void_p(wp.data_ptr()), _chk(shift, "shift"),
                                              Co, H, W, _chk(out, "out"), ctypes.c_void_p(ga[0].data_ptr()), int(ga[1]), _stream())
            if rc != _lib.ERR_UNSUPPORTED:
                _lib.check(rc, "d3d_conv2d_k3_zs_h16_gn")
                dispatch_counts["conv2d_tile"] += 1
                dispatch_counts["conv2d_gn_fused"] += 1
                return out
            gn.slot = None   # (not taken: nothing was launched, the slot stays zero for the next request of this lap)
            _gn_arenas[(x.device.index, torch.cuda.current_stream(x.device).cuda_stream)][1] -= 1
    if bf16:
        name, wp = "d3d_conv2d_k3_zs_h16", derived_weight(weight, "z2bf16", _pack_z2_bf16)
    elif _z2_fp32_entry() == "x3" and Ci != 48:   # (split cells of 48 channels + their weights do not fit the LDS)
        name, wp = "d3d_conv2d_k3_zs_bf16x3", derived_weight(weight, "z2bf16x3", _pack_z2_bf16x3)
    else:
        name, wp = "d3d_conv2d_k3_zs_f32", derived_weight(weight, "z2f32", _pack_z2_f32)
    out = torch.empty((Co, H, W), dtype=torch.float32, device=x.device)
    rc = getattr(lib, name)(_chk(x, "x", 3), Ci0, _opt(x2, "x2"), Ci1, ctypes.c_void_p(wp.data_ptr()), _opt(scale, "scale"),
                            _opt(shift, "shift"), _opt(skip, "skip"), _opt(aux1, "aux1"), int(act), int(ep_split),
                            int(bool(skip_after_act)), Co, H, W, _chk(out, "out"), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, name)
    dispatch_counts["conv2d_tile"] += 1
    return out


def conv2d_k3_pair3(x, w0, scale0, shift0, act0, w1, scale1, shift1, act1):
    """conv0 of a feature trunk -- Conv3x3(3 -> 8) + affine + act0, then Conv3x3(8 -> 8) + affine + act1 at full resolution
    (module.py:663-666) -- in ONE launch of the tile kernel: the 8-channel map between the layers never reaches memory
    (d3d_conv2d_k3_pair3_bf16x3; bit for bit what conv2d_stream followed by conv2d_zs give, which is what conv2d_k3 runs for the
    two layers in the fp32 precision of the feature nets at these sizes).  x [3,H,W].  None when it does not apply."""
    if x.dim() != 3 or x.shape[0] != 3 or conv_precision() == "h16" or _z2_fp32_entry() != "x3" or not _use_mfma() \
            or _cfg.off("conv0_pair") or _cfg.off("conv2d_zs") or _cfg.off("conv2d_stream"):
        return None
    _, H, W = x.shape
    if tuple(w0.shape) != (8, 3, 3, 3) or tuple(w1.shape) != (8, 8, 3, 3) or W % 4 or H * W < _CONV2D_STREAM_MIN \
            or 8 * H * W * 4 >= 2 ** 31 or act0 not in (0, 1) or act1 not in (0, 1):
        return None

    def pack0(w):
        wp = w.new_zeros((8, 3, 3, 8))
        wp[:3] = w.permute(1, 2, 3, 0)
        return wp
    wp0 = derived_weight(w0, "c2s", pack0)   # (the packing of conv2d_stream: same key, same tensor)
    wp1 = derived_weight(w1, "z2bf16x3", _pack_z2_bf16x3)
    out = torch.empty((8, H, W), dtype=torch.float32, device=x.device)
    rc = _lib.load().d3d_conv2d_k3_pair3_bf16x3(_chk(x, "x", 3), _chk(wp0, "w0packed"), _opt(scale0, "scale0"), _opt(shift0, "shift0"),
                                                int(act0), ctypes.c_void_p(wp1.data_ptr()), _opt(scale1, "scale1"),
                                                _opt(shift1, "shift1"), int(act1), 8, H, W, _chk(out, "out"), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "d3d_conv2d_k3_pair3_bf16x3")
    dispatch_counts["conv2d_pair3"] += 1
    return out


# In the default fp32 precision the tile kernels serve the slice regularisers only (the feature pyramids keep their tuned
# vector-unit kernels: a FeatureNet forward is 2.12 ms on those, 2.19 ms on the fp32 tile kernel): the regulariser
# modules switch them on around their forward.


class slice_tile_kernels:
    """Context manager: stride-2 / transposed 2-D layers inside may use the fp32 tile kernels."""

    def __enter__(self):
        self.saved = _cfg.state.tile_kernels
        _cfg.state.tile_kernels = True

    def __exit__(self, *exc):
        _cfg.state.tile_kernels = self.saved


def conv2d_s2_zs(x, weight, scale=None, shift=None, skip=None, act=0, skip_after_act=False):
    """3x3 stride-2 conv (pad 1) with bf16 matrix-core operands on the tile kernel; None for shapes it does not take."""
    Ci, H, W = x.shape
    Co = weight.shape[0]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    bf16 = conv_precision() == "h16"
    x3 = not bf16 and _z2_fp32_entry() == "x3" and Ci != 48
    # the stride-1 kernel with a subsampled store: the pair-visibility UNet's 48 channels, RED-Net's 32 -> 64 layer (fast mode)
    wide = ((Ci == 48 and Co <= 48) or (bf16 and Ci == 32 and Co <= 64)) and W % 4 == 0
    if not wide and (Ci not in ((8, 16) if bf16 or x3 else (8,)) or Co > 32 or Wo % 4) or act not in (0, 1) \
            or _cfg.off("conv2d_zs"):
        return None
    if tuple(weight.shape) != (Co, Ci, 3, 3):
        raise ValueError("weight must be [Co,%d,3,3] (got %s)" % (Ci, tuple(weight.shape)))
    if bf16:
        fn, wp = _lib.load().d3d_conv2d_k3s2_zs_h16, derived_weight(weight, "z2bf16", _pack_z2_bf16)
    elif x3:
        fn, wp = _lib.load().d3d_conv2d_k3s2_zs_bf16x3, derived_weight(weight, "z2bf16x3", _pack_z2_bf16x3)
    else:
        fn, wp = _lib.load().d3d_conv2d_k3s2_zs_f32, derived_weight(weight, "z2f32", _pack_z2_f32)
    out = torch.empty((Co, Ho, Wo), dtype=torch.float32, device=x.device)
    if skip is not None and skip.shape != out.shape:
        raise ValueError("skip shape mismatch")
    rc = fn(_chk(x, "x", 3), ctypes.c_void_p(wp.data_ptr()), _opt(scale, "scale"), _opt(shift, "shift"), _opt(skip, "skip"),
            int(act), int(bool(skip_after_act)), Ci, Co, H, W, _chk(out, "out"), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "d3d_conv2d_k3s2_zs")
    return out


def conv2d_s2_zs_batched(x, weight, act=1):
    """3 x 3 stride-2 conv (pad 1, no bias) over a BATCH of maps x [B,Ci,H,W] in one launch of the tile kernel (h16 mode:
    d3d_conv2d_k3s2_zs_h16_batched; RED-Net's encoder for every depth slice of a stage, msrednet.py:352-356).  Per item bit for
    bit conv2d_s2_zs.  Returns [B,Co,Ho,Wo], or None for shapes / modes the kernel does not take."""
    if x.dim() != 4 or conv_precision() != "h16" or _cfg.off("conv2d_zs") or act not in (0, 1):
        return None
    B, Ci, H, W = x.shape
    Co = weight.shape[0]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    wide = Ci == 32 and Co <= 64 and W % 4 == 0
    if not wide and (Ci not in (8, 16) or Co > 32 or Wo % 4):
        return None
    if tuple(weight.shape) != (Co, Ci, 3, 3):
        raise ValueError("weight must be [Co,%d,3,3] (got %s)" % (Ci, tuple(weight.shape)))
    wp = derived_weight(weight, "z2bf16", _pack_z2_bf16)
    out = torch.empty((B, Co, Ho, Wo), dtype=torch.float32, device=x.device)
    rc = _lib.load().d3d_conv2d_k3s2_zs_h16_batched(_chk(x, "x", 4), ctypes.c_void_p(wp.data_ptr()), None, None, int(act), Ci, Co, H, W, B,
                                                     Ci * H * W, Co * Ho * Wo, _chk(out, "out"), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "d3d_conv2d_k3s2_zs_h16_batched")
    dispatch_counts["conv2d_s2_batched"] += 1
    return out


def _pack_t2d_bf16(w, dt=None):
    """nn.ConvTranspose2d weight [Ci,Co,3,3] -> B operands for d3d_convtranspose2d_k3s2_zs_h16: per output parity class
    (py,px), order py*2 + px, taps (dy,dx) with d <= p per dimension, dy-major; an even output coordinate uses kernel index 1
    (d = 0), an odd one index 2 (d = 0) and 0 (d = 1).  K = (tap, ci) padded to 32, 16 output columns; [K block][lane][8]."""
    Ci, Co = w.shape[0], w.shape[1]
    kmap = {(0, 0): 1, (1, 0): 2, (1, 1): 0}
    parts = []
    for c in range(4):
        py, px = c >> 1, c & 1
        taps = [(dy, dx) for dy in range(1 + py) for dx in range(1 + px)]
        K = len(taps) * Ci
        nkb = (K + 31) // 32
        b = torch.zeros((nkb * 32, 16), dtype=torch.float32, device=w.device)
        for t, (dy, dx) in enumerate(taps):
            b[t * Ci:(t + 1) * Ci, :Co] = w[:, :, kmap[(py, dy)], kmap[(px, dx)]]
        b = b.reshape(nkb, 4, 8, 16).permute(0, 1, 3, 2)                 # [kb][kgroup][n][j]
        parts.append(b.reshape(nkb * 64, 8))
    return torch.cat(parts).to(dt or h16_dtype()).view(torch.int16).contiguous()


def _pack_t2d_f32(w):
    """_pack_t2d_bf16 in fp32 for d3d_convtranspose2d_k3s2_zs_f32: per parity class, K = (tap, ci) in blocks of 4, [K block][lane]
    with lane l = column l & 15, K row l >> 4."""
    Ci, Co = w.shape[0], w.shape[1]
    kmap = {(0, 0): 1, (1, 0): 2, (1, 1): 0}
    parts = []
    for c in range(4):
        py, px = c >> 1, c & 1
        taps = [(dy, dx) for dy in range(1 + py) for dx in range(1 + px)]
        K = len(taps) * Ci
        b = torch.zeros((K, 16), dtype=torch.float32, device=w.device)
        for t, (dy, dx) in enumerate(taps):
            b[t * Ci:(t + 1) * Ci, :Co] = w[:, :, kmap[(py, dy)], kmap[(px, dx)]]
        parts.append(b.reshape(K // 4, 64))
    return torch.cat(parts).contiguous()


def convtranspose2d_zs(x, weight, scale=None, shift=None, skip=None, act=0, skip_after_act=False):
    """ConvTranspose2d(k 3, stride 2, pad 1, output_pad 1) with bf16 matrix-core operands on the tile kernel (four per-parity
    convolutions over one staged patch); None for shapes it does not take."""
    Ci, H, W = x.shape
    Co = weight.shape[1]
    if Ci not in (8, 16, 32) or Co > 16 or W % 4 or act not in (0, 1) or _cfg.off("conv2d_zs"):
        return None
    if tuple(weight.shape) != (Ci, Co, 3, 3):
        raise ValueError("weight must be [%d,Co,3,3] (got %s)" % (Ci, tuple(weight.shape)))
    if conv_precision() == "h16":
        fn, wp = _lib.load().d3d_convtranspose2d_k3s2_zs_h16, derived_weight(weight, "t2dbf16", _pack_t2d_bf16)
    elif _z2_fp32_entry() == "x3":
        fn, wp = _lib.load().d3d_convtranspose2d_k3s2_zs_bf16x3, derived_weight(weight, "t2dbf16x3", _pack_t2d_bf16x3)
    else:
        fn, wp = _lib.load().d3d_convtranspose2d_k3s2_zs_f32, derived_weight(weight, "t2df32", _pack_t2d_f32)
    out = torch.empty((Co, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    if skip is not None and skip.shape != out.shape:
        raise ValueError("skip shape mismatch")
    rc = fn(_chk(x, "x", 3), ctypes.c_void_p(wp.data_ptr()), _opt(scale, "scale"), _opt(shift, "shift"), _opt(skip, "skip"),
            int(act), int(bool(skip_after_act)), Ci, Co, H, W, _chk(out, "out"), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "d3d_convtranspose2d_k3s2_zs")
    dispatch_counts["convtranspose2d_tile"] += 1
    return out


def conv2d_wide(x, weight, scale=None, shift=None, skip=None, act=0, x2=None, gn=None):
    """3x3 stride-1 conv over cat(x, x2) with 64 | 128 input channels (parts of 32) and 32 | 64 | 128 output channels on the
    bf16 matrix cores, K walked in chunks of 32 channels (d3d_conv2d_k3_wide_h16, csrc/conv2d_wide.hip: the coarse conv-GRU
    levels of the RED-Net slice regulariser).  bf16 mode only; None for other shapes."""
    Ci0, H, W = x.shape
    Ci1 = 0 if x2 is None else x2.shape[0]
    Co = weight.shape[0]
    if conv_precision() != "h16" or _cfg.off("conv2d_wide") or (Ci0 + Ci1) not in (64, 128) or Ci0 % 32 or Ci1 % 32 \
            or Co not in (32, 64, 128) or act not in (0, 1) or not _use_mfma():
        return None
    wp = derived_weight(weight, "z2bf16", _pack_z2_bf16)
    out = torch.empty((Co, H, W), dtype=torch.float32, device=x.device)
    ga = _gn_args(gn, Co, act, scale, skip, x.device) if shift is not None else None
    if ga is not None:   # the layer and the GroupNorm statistics of its output in one launch
        rc = _lib.load().d3d_conv2d_k3_wide_h16_gn(_chk(x, "x", 3), Ci0, _opt(x2, "x2"), Ci1, ctypes.c_void_p(wp.data_ptr()),
                                                    _chk(shift, "shift"), Co, H, W, _chk(out, "out"), ctypes.c_void_p(ga[0].data_ptr()),
                                                    int(ga[1]), _stream())
        if rc != _lib.ERR_UNSUPPORTED:
            _lib.check(rc, "d3d_conv2d_k3_wide_h16_gn")
            dispatch_counts["conv2d_wide"] += 1
            dispatch_counts["conv2d_gn_fused"] += 1
            return out
        gn.slot = None
        _gn_arenas[(x.device.index, torch.cuda.current_stream(x.device).cuda_stream)][1] -= 1
    rc = _lib.load().d3d_conv2d_k3_wide_h16(_chk(x, "x", 3), Ci0, _opt(x2, "x2"), Ci1, ctypes.c_void_p(wp.data_ptr()),
                                             _opt(scale, "scale"), _opt(shift, "shift"), _opt(skip, "skip"), int(act), Co, H, W,
                                             _chk(out, "out"), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "d3d_conv2d_k3_wide_h16")
    dispatch_counts["conv2d_wide"] += 1
    return out


def conv2d_k3(x, weight, scale=None, shift=None, skip=None, act=0, stride=1, x2=None, gn=None):
    """3x3 conv over cat(x, x2) channels. x [Ci0,H,W], x2 [Ci1,H,W]|None, weight [Co,Ci0+Ci1,3,3].  gn: an optional GnStats request
    for the GroupNorm statistics of the output, served in the layer's epilogue where the kernel has that form."""
    Ci0, H, W = x.shape
    Ci1 = 0 if x2 is None else x2.shape[0]
    Co = weight.shape[0]
    if tuple(weight.shape) != (Co, Ci0 + Ci1, 3, 3):
        raise ValueError("weight must be [Co,%d,3,3] (got %s)" % (Ci0 + Ci1, tuple(weight.shape)))
    if x2 is not None and tuple(x2.shape[1:]) != (H, W):
        raise ValueError("x2 spatial size mismatch")
    if stride == 1 and act in (0, 1) and _use_mfma() and H * W >= CONV2D_ZS_MINPIX and \
            (conv_precision() == "h16" or _cfg.state.tile_kernels):
        # bf16 mode, and the ConvReLU of a slice regulariser in fp32 mode (three-way bf16 splits: 65.2 -> 63.9 ms per AdaMVS
        # view; on the fp32 instruction the vector-unit kernel won, 71.7 vs 73.1 ms): one tile per step on the matrix cores
        y = conv2d_zs(x, weight, scale, shift, skip, act, x2=x2, skip_after_act=True, gn=gn)   # conv2d_k3: the skip is added last
        if y is not None:
            return y
    if stride == 1 and act in (0, 1) and (Ci0 + Ci1) in (64, 128):
        y = conv2d_wide(x, weight, scale, shift, skip, act, x2=x2, gn=gn)   # (bf16 mode: the wide conv-GRU levels of RED-Net)
        if y is not None:
            return y
    zs_any = conv_precision() == "h16" or _cfg.state.tile_kernels or Ci0 == 48
    if stride == 2 and x2 is None and act in (0, 1) and zs_any and _use_mfma() and H * W >= (64 * 64 if Ci0 == 48 else 128 * 128):
        y = conv2d_s2_zs(x, weight, scale, shift, skip, act, skip_after_act=True)
        if y is not None:
            return y
    # (the concat convolutions of AdaMVS's pyramid decoder, adamvs.py:104-113: 16 -> 8 at full, 32 -> 16 at half resolution --
    #  228 / 148 us on the tile kernel against 272 / 194 us on the vector-unit kernel since the round-4 rebuild of the tile kernels)
    fuse = x2 is not None and (Ci0 + Ci1, Co) in ((16, 8), (32, 16))
    if stride == 1 and act in (0, 1) and ((x2 is None and (Ci0, Co) in ((32, 32), (16, 16), (8, 8), (48, 48))) or fuse) and _use_mfma() \
            and H * W >= (64 * 64 if Ci0 == 48 else 128 * 128):
        # 32 -> 32, 16 -> 16 and 8 -> 8 (the trunks of the feature pyramids) in fp32 accuracy: the tile kernel beats the
        # row-streamed matrix-core form (140 -> 68 us at 464 x 688) and the vector-unit kernel (140 -> 68 us at 928 x 1376,
        # 180 -> 130 us at 1856 x 2752); 32 -> 8 | 16 (the FPN output layers) lose there and stay on the kernels below.
        # 48 -> 48 (the pair-visibility UNet of AdaMVS): fp32 instruction, one patch buffer (340 -> 90 us at 688 x 464)
        y = conv2d_zs(x, weight, scale, shift, skip, act, x2=x2, skip_after_act=True)
        if y is not None:
            return y
    if stride == 1 and act in (0, 1):
        y = conv2d_stream(x, weight, scale, shift, skip, act, x2=x2)
        if y is not None:
            return y
    if _use_mfma() and Co <= 64:
        y = conv_k3_mfma(x, weight, scale, shift, skip, act=act, stride=stride, x2=x2)
        if y is not None:
            return y
    if _use_mfma() and Co > 64 and Co % 64 == 0:
        # wide layers (RED-Net's 128-channel gate convolution): 64 output channels per matrix-core launch
        parts = []
        for c0 in range(0, Co, 64):
            wsl = derived_weight(weight, "rows%d" % c0, lambda w, c0=c0: w[c0:c0 + 64])
            sl = lambda t: None if t is None else t[c0:c0 + 64].contiguous()
            parts.append(conv_k3_mfma(x, wsl, sl(scale), sl(shift), sl(skip), act=act, stride=stride, x2=x2))
        if all(q is not None for q in parts):
            return torch.cat(parts, 0)
    o = lambda n: (n - 1) // stride + 1
    out = torch.empty((Co, o(H), o(W)), dtype=torch.float32, device=x.device)
    if skip is not None and skip.shape != out.shape:
        raise ValueError("skip shape mismatch")
    rc = _lib.load().d3d_conv2d_k3(_chk(x, "x", 3), Ci0, _opt(x2, "x2"), Ci1, _chk(weight, "weight"),
                                   _opt(scale, "scale"), _opt(shift, "shift"), _opt(skip, "skip"), int(act), Co, H,
                                   W, stride, _chk(out, "out"), _stream())
    _lib.check(rc, "d3d_conv2d_k3")
    return out


def convtranspose2d_k3s2(x, weight, scale=None, shift=None, skip=None, skip_after_act=False, act=0):
    """x [Ci,H,W], weight [Ci,Co,3,3] -> [Co,2H,2W]."""
    Ci, H, W = x.shape
    Co = weight.shape[1]
    if tuple(weight.shape) != (Ci, Co, 3, 3):
        raise ValueError("weight must be [%d,Co,3,3] (got %s)" % (Ci, tuple(weight.shape)))
    # (fp32 mode: the tile kernel serves the slice regularisers and the feature pyramids' deconvs alike)
    if _use_mfma() and act in (0, 1) and H * W >= 64 * 64:
        y = convtranspose2d_zs(x, weight, scale, shift, skip, act=act, skip_after_act=skip_after_act)
        if y is not None:
            return y
    if Ci == 48 and Co <= 48 and _use_mfma() and act in (0, 1) and W % 2 == 0 and H * W <= 512 * 512:
        # the pair-visibility UNet (adamvs.py:198-238): no transposed tile kernel has room for 48-channel cells, and at its
        # image sizes the four per-parity launches of round 1's kernel are latency, not work.  A transposed convolution (k 3,
        # s 2, p 1, output_pad 1) is the stride-1 convolution of the zero-stuffed input with the flipped kernel: one launch
        # of the tile kernel (three quarters of its products are zeros)
        z = x.new_zeros((Ci, 2 * H, 2 * W))
        z[:, ::2, ::2] = x
        wf = derived_weight(weight, "t2flip", lambda w: w.flip(2, 3).transpose(0, 1))
        y = conv2d_zs(z, wf, scale, shift, skip, act, skip_after_act=skip_after_act)
        if y is not None:
            return y
    if Ci == 64 and Co in (32, 64) and conv_precision() == "h16" and _use_mfma() and act in (0, 1) and (skip is None or skip_after_act) \
            and not _cfg.off("convt_wide") and H * W <= 512 * 512:
        # RED-Net's upconv3 (msrednet.py:348: ConvTransReLU(64, 32) from the coarsest level, 86 x 58 pixels at stage 1): the same
        # zero-stuffed form on the wide tile kernel (64 input channels in two chunks of 32; ReLU, then the skip: ConvTransReLU's
        # order).  167 / 102 us per call on the round-1 stream kernel -- 12.6 ms of a 66 ms view
        # -- the stuffed image formed in the kernel's staging: no fill + strided copy of a [64,2H,2W] tensor per call (two of a
        # slice's 33 launches; stage 1 replays its captured loop at ~9 us per node)
        wf = derived_weight(weight, "t2flip", lambda w: w.flip(2, 3).transpose(0, 1))
        wp = derived_weight(wf, "z2bf16", _pack_z2_bf16)
        if skip is not None and tuple(skip.shape) != (Co, 2 * H, 2 * W):
            raise ValueError("skip shape mismatch")
        out = torch.empty((Co, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
        rc = _lib.load().d3d_convtranspose2d_k3s2_wide_h16(_chk(x, "x", 3), ctypes.c_void_p(wp.data_ptr()), _opt(scale, "scale"),
                                                            _opt(shift, "shift"), _opt(skip, "skip"), int(act), Ci, Co, H, W,
                                                            _chk(out, "out"), _stream())
        if rc != _lib.ERR_UNSUPPORTED:
            _lib.check(rc, "d3d_convtranspose2d_k3s2_wide_h16")
            dispatch_counts["convtranspose2d_wide"] += 1
            return out
    if _use_mfma() and Co <= 64:
        y = convtranspose_k3s2_mfma(x, weight, scale, shift, skip, act=act, skip_after_act=skip_after_act)
        if y is not None:
            return y
    out = torch.empty((Co, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    if skip is not None and skip.shape != out.shape:
        raise ValueError("skip shape mismatch")
    rc = _lib.load().d3d_convtranspose2d_k3s2(_chk(x, "x", 3), _chk(weight, "weight"), _opt(scale, "scale"),
                                              _opt(shift, "shift"), _opt(skip, "skip"), int(skip_after_act),
                                              int(act), Ci, Co, H, W, _chk(out, "out"), _stream())
    _lib.check(rc, "d3d_convtranspose2d_k3s2")
    return out


def gru_gates(gates, h):
    """gates [2Hc,H,W] (pre-activation), h [Hc,H,W] -> (r*h, u)."""
    Hc = h.shape[0]
    plane = h.shape[1] * h.shape[2]
    if gates.shape[0] != 2 * Hc or tuple(gates.shape[1:]) != tuple(h.shape[1:]):
        raise ValueError("gates must be [2*%d,H,W]" % Hc)
    rh = torch.empty_like(h)
    u = torch.empty_like(h)
    rc = _lib.load().d3d_gru_gates(_chk(gates, "gates"), _chk(h, "h"), Hc, plane, _chk(rh, "rh"), _chk(u, "u"),
                                   _stream())
    _lib.check(rc, "d3d_gru_gates")
    return rh, u


def gru_update(u, h, convc):
    out = torch.empty_like(h)
    rc = _lib.load().d3d_gru_update(_chk(u, "u"), _chk(h, "h"), _chk(convc, "convc"), h.numel(), _chk(out, "out"),
                                    _stream())
    _lib.check(rc, "d3d_gru_update")
    return out


_gn_arenas = {}
_GN_SLOTS = 2048


class GnStats:
    """Request for the GroupNorm(1, C) statistics of a convolution's output (ConvGRUCell2, module.py:62-99): `ngroups` (1 | 2) equal
    consecutive channel groups.  conv2d_k3(..., gn=req) fills `req` in the convolution's epilogue where the kernel has that form
    (bf16 mode: d3d_conv2d_k3_zs_h16_gn / d3d_conv2d_k3_wide_h16_gn); `req.stats(y)` returns the fp64 (sum, sum of squares) pairs
    -- the epilogue's, or those of d3d_groupnorm_stats over the stored tensor y.  The pairs live in a zeroed arena of slots per
    (device, stream): one fill per _GN_SLOTS requests instead of one per layer."""

    def __init__(self, ngroups):
        self.ngroups, self.slot = int(ngroups), None

    def take_slot(self, device):
        key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        ar = _gn_arenas.get(key)
        if ar is None or ar[1] >= _GN_SLOTS:
            buf = ar[0] if ar is not None else torch.empty((_GN_SLOTS, 2, 2), dtype=torch.float64, device=device)
            buf.zero_()   # (stream order: behind every kernel that used the slots of the previous lap)
            ar = [buf, 0]
            _gn_arenas[key] = ar
        self.slot = ar[0][ar[1]]
        ar[1] += 1
        return self.slot

    def stats(self, y):
        if self.slot is None:
            return groupnorm_stats(y, self.ngroups)
        return self.slot[0] if self.ngroups == 1 else self.slot


def _gn_args(gn, Co, act, scale, skip, device):
    """(stats pointer, split) for the _gn entries, or None when the request cannot ride on this layer."""
    if gn is None or _cfg.off("gn_fused") or act != 0 or scale is not None or skip is not None or (gn.ngroups == 2 and Co % 2):
        return None
    return gn.take_slot(device), (Co // 2 if gn.ngroups == 2 else Co)


def groupnorm_stats(x, ngroups=1):
    """(sum, sum of squares) of each of `ngroups` equal consecutive parts of x as device fp64 pairs [ngroups,2],
    for GroupNorm(1, C) (module.py:62-67); ngroups = 1 returns the single pair [2]."""
    st = torch.empty((ngroups, 2), dtype=torch.float64, device=x.device)
    rc = _lib.load().d3d_groupnorm_stats(_chk(x, "x"), x.numel() // ngroups, ngroups, ctypes.c_void_p(st.data_ptr()),
                                         _stream())
    _lib.check(rc, "d3d_groupnorm_stats")
    return st[0] if ngroups == 1 else st


def gru_reset_gn(gates, h, gamma_r, beta_r, eps, stats_r):
    """The reset half of ConvGRUCell2's gates alone (module.py:71-76,85): rh = sigmoid(GroupNorm(gates[:Hc])) * h.  None when the
    kernel does not take the shape (the caller then runs gru_gates_gn)."""
    Hc, plane = h.shape[0], h[0].numel()
    rh = torch.empty_like(h)
    rc = _lib.load().d3d_gru_reset_gn(_chk(gates, "gates"), _dptr(stats_r), _chk(gamma_r, "gamma_r"), _chk(beta_r, "beta_r"), _chk(h, "h"),
                                      Hc, plane, float(eps), int(conv_precision() == "h16"), _chk(rh, "rh"), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "d3d_gru_reset_gn")
    return rh


def gru_update_gates_gn(o, gates, h, gamma, beta, gamma_u, beta_u, eps, stats_o, stats_u):
    """ConvGRUCell2's state update with the update gate evaluated in place (module.py:77-82,84-98): u = sigmoid(GroupNorm(gates[Hc:])),
    h' = u*h + (1-u)*tanh(GroupNorm(o)) -- gru_gates_gn's u followed by gru_update_gn, bit for bit, without u's write and read."""
    Hc, plane = h.shape[0], h[0].numel()
    out = torch.empty_like(h)
    rc = _lib.load().d3d_gru_update_gates_gn(_chk(o, "o"), _dptr(stats_o), _chk(gamma, "gamma"), _chk(beta, "beta"), _chk(gates, "gates"),
                                             _dptr(stats_u), _chk(gamma_u, "gamma_u"), _chk(beta_u, "beta_u"), _chk(h, "h"), Hc, plane,
                                             float(eps), int(conv_precision() == "h16"), _chk(out, "out"), _stream())
    _lib.check(rc, "d3d_gru_update_gates_gn")
    return out


_gru2_scratch = {}


def gru2_cell_gn(x, h, w_gates, b_gates, w_cand, b_cand, norm_r, norm_u, norm_o):
    """One ConvGRUCell2 step (module.py:53-99) as ONE library call (d3d_gru2_cell_gn_h16): the cell's four launches issued back to back
    from C -- the gate / candidate convolutions with the GroupNorm statistics in their epilogues, the reset pass, the state update
    with the update gate evaluated in place: the kernels and operands of conv2d_k3(gn=) + gru_reset_gn + conv2d_k3(gn=) +
    gru_update_gates_gn, without the host's work between them.  The intermediates (pre-norm gates, r*h, pre-norm candidate) live in a
    scratch set per (stream, shape), reused from cell to cell in stream order.  norm_*: the three nn.GroupNorm(1, C) modules.
    h16 mode; returns the new state, or None where the entry point does not take the shapes (the caller runs the layers)."""
    if conv_precision() != "h16" or not _use_mfma() or _cfg.off("gru2_cell") or _cfg.off("gn_fused") or _cfg.off("gru_gates_split") \
            or _cfg.off("conv2d_zs") or _cfg.off("conv2d_wide") or b_gates is None or b_cand is None:
        return None
    Cx, H, W = x.shape
    Hc = h.shape[0]
    if not (norm_r.eps == norm_u.eps == norm_o.eps) or tuple(h.shape[1:]) != (H, W) or (H * W) % 4:
        return None
    Ci = Cx + Hc
    if not ((Ci in (16, 24, 32, 40) and W % 4 == 0 and Cx % 8 == 0 and 2 * Hc <= (16 if Ci in (24, 40) else 32))
            or (Ci in (64, 128) and Cx % 32 == 0 and Hc in (32, 64))):
        return None
    dev = x.device
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, Hc, H, W)
    sc = _gru2_scratch.get(key)
    if sc is None:
        if len(_gru2_scratch) > 256:
            _gru2_scratch.clear()
        sc = _gru2_scratch[key] = (torch.empty((2 * Hc, H, W), dtype=torch.float32, device=dev), torch.empty((Hc, H, W), dtype=torch.float32, device=dev),
                                   torch.empty((Hc, H, W), dtype=torch.float32, device=dev))
    wg = derived_weight(w_gates, "z2bf16", _pack_z2_bf16)
    wc = derived_weight(w_cand, "z2bf16", _pack_z2_bf16)
    sg, so = GnStats(2).take_slot(dev), GnStats(1).take_slot(dev)
    out = torch.empty_like(h)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = _lib.load().d3d_gru2_cell_gn_h16(_chk(x, "x", 3), Cx, _chk(h, "h", 3), Hc, H, W, p(wg), _chk(b_gates, "b_gates"), p(wc), _chk(b_cand, "b_cand"),
                                          _chk(norm_r.weight, "gamma_r"), _chk(norm_r.bias, "beta_r"), _chk(norm_u.weight, "gamma_u"),
                                          _chk(norm_u.bias, "beta_u"), _chk(norm_o.weight, "gamma_o"), _chk(norm_o.bias, "beta_o"),
                                          float(norm_r.eps), 1, p(sg), p(so), p(sc[0]), p(sc[1]), p(sc[2]), p(out), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        _gn_arenas[(dev.index, torch.cuda.current_stream(dev).cuda_stream)][1] -= 2   # (nothing was launched: the two slots stay zero)
        return None
    _lib.check(rc, "d3d_gru2_cell_gn_h16")
    dispatch_counts["gru2_cell"] += 1
    dispatch_counts["conv2d_gn_fused"] += 2
    dispatch_counts["conv2d_wide" if Ci >= 64 else "conv2d_tile"] += 2
    return out


def _dptr(t):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.numel() == 2):
        raise TypeError("statistics must be a CUDA float64 pair")
    return ctypes.c_void_p(t.data_ptr())


def gru_gates_gn(gates, h, gamma_r, beta_r, gamma_u, beta_u, eps=1e-5, stats=None):
    """ConvGRUCell2 gates (module.py:71-82): gates [2Hc,H,W] pre-norm, h [Hc,H,W] -> (r*h, u)."""
    Hc = h.shape[0]
    plane = h[0].numel()
    if gates.shape[0] != 2 * Hc or gates[0].numel() != plane:
        raise ValueError("gates must be [2*Hc,H,W]")
    st = groupnorm_stats(gates, 2) if stats is None else stats  # reset-gate half, update-gate half: one launch (or the convolution's own)
    st_r, st_u = st[0], st[1]
    rh = torch.empty_like(h)
    u = torch.empty_like(h)
    rc = _lib.load().d3d_gru_gates_gn(_chk(gates, "gates"), _dptr(st_r), _dptr(st_u), _chk(gamma_r, "gamma_r"),
                                      _chk(beta_r, "beta_r"), _chk(gamma_u, "gamma_u"), _chk(beta_u, "beta_u"),
                                      _chk(h, "h"), Hc, plane, float(eps), int(conv_precision() == "h16"), _chk(rh, "rh"), _chk(u, "u"), _stream())
    _lib.check(rc, "d3d_gru_gates_gn")
    return rh, u


def gru_update_gn(o, u, h, gamma, beta, eps=1e-5, stats=None):
    """ConvGRUCell2 state update (module.py:84-98): h' = u*h + (1-u)*tanh(GroupNorm(o))."""
    Hc = h.shape[0]
    plane = h[0].numel()
    st = groupnorm_stats(o) if stats is None else stats
    out = torch.empty_like(h)
    rc = _lib.load().d3d_gru_update_gn(_chk(o, "o"), _dptr(st), _chk(gamma, "gamma"), _chk(beta, "beta"), _chk(u, "u"),
                                       _chk(h, "h"), Hc, plane, float(eps), int(conv_precision() == "h16"), _chk(out, "out"), _stream())
    _lib.check(rc, "d3d_gru_update_gn")
    return out


_derived_cache = {}


_side_streams = {}   # (device index, caller stream, owner) -> side streams
_side_lock = __import__("threading").Lock()


def side_streams(device, n, owner="ops"):
    """`n` side streams for the forward that runs on the CALLER'S CURRENT stream of `device` -- one set per (device, caller stream,
    owner), created on first use.  Two forwards in flight on different streams (two host threads) therefore never
    share a side stream: their forks / joins do not serialise on each other, the GroupNorm slot arenas (keyed by stream) are
    not shared, and a block the caching allocator frees on a side stream is reused behind THAT caller's next fork only."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, owner)
    with _side_lock:
        side = _side_streams.get(key)
        if side is None or len(side) < n:
            side = _side_streams[key] = [torch.cuda.Stream(device) for _ in range(n)]
    return side[:n]


def hand_over(outs, stream):
    """Tensors produced on a side stream and consumed on `stream` from now on: tell the caching allocator (record_stream), so
    that their blocks -- allocated in the side stream's pool -- are not handed out again on the side stream while `stream` still
    reads them.  Walks lists / tuples / dicts.  (Ordering is the join event's job; this is the allocator's bookkeeping.)"""
    if isinstance(outs, torch.Tensor):
        if outs.is_cuda and not _cfg.off("hand_over"):
            outs.record_stream(stream)
    elif isinstance(outs, (list, tuple)):
        for o in outs:
            hand_over(o, stream)
    elif isinstance(outs, dict):
        for o in outs.values():
            hand_over(o, stream)


def on_streams(thunks, device, switch):
    """[f() for f in thunks] with the INDEPENDENT pieces of work going round-robin over the caller's stream and two side streams
    (fork event before, one join event per side piece after): chains of small launches that leave most of the chip idle overlap.
    Same kernels, same operands; `switch` (a key of config.KERNELS) in D3D_KERNELS_OFF keeps everything on the caller's stream.
    The side streams belong to the caller's stream (side_streams), and what the side pieces return is handed over to it."""
    if len(thunks) < 2 or device.type != "cuda" or _cfg.off(switch):
        return [f() for f in thunks]
    main = torch.cuda.current_stream(device)
    side = side_streams(device, 2)
    fork = main.record_event()
    outs, joins = [], []
    for i, f in enumerate(thunks):
        st = (None, side[0], side[1])[i % 3]
        if st is None:
            outs.append(f())
            continue
        with torch.cuda.stream(st):
            st.wait_event(fork)
            outs.append(f())
            joins.append(st.record_event())
        hand_over(outs[-1], main)
    for e in joins:
        main.wait_event(e)
    return outs


# The drivers need the depth range (depth_values[0, 0], depth_values[0, -1]) as host numbers (adamvs.py:565-566 and its siblings
# read it with .item()): on a device tensor that is a device -> host copy, i.e. the host waits for every kernel of the PREVIOUS
# view before it launches the first one of this view.  A caller that built the tensor from host data says so once
# (note_depth_range: predict_views, bench.py) and the forward then never touches the device for it.
_depth_ranges = {}


def note_depth_range(depth_values, dmin, dmax):
    """`depth_values` (a device tensor about to be passed to an Infer_* forward) holds [dmin .. dmax] in its first row: keep the
    host copy of the two numbers (until the tensor is written to or dies)."""
    import weakref

    key = id(depth_values)
    ref = weakref.ref(depth_values, lambda _r, key=key: _depth_ranges.pop(key, None))
    _depth_ranges[key] = (ref, depth_values._version, float(dmin), float(dmax))
    return depth_values


def depth_range_host(depth_values):
    """(dmin, dmax) of an Infer_* forward's depth_values [B,2] | [B,D] as host floats: the noted pair if the caller left one
    (no device access), else read from the tensor (one host sync)."""
    hit = _depth_ranges.get(id(depth_values))
    if hit is not None and hit[0]() is depth_values and hit[1] == depth_values._version:
        return hit[2], hit[3]
    dmin, dmax = (float(v) for v in depth_values[0, [0, -1]].tolist())
    return dmin, dmax


def publish_prepared(weight):
    """A freshly prepared (packed / folded) operand goes into a cache that EVERY stream reads: the forwards run some layers on
    side streams (feature pyramids, RED-Net's conv-GRU levels), so the stream that prepared it waits for the preparation once --
    a cache miss happens at the first forward after a weight changes -- and whoever finds the entry later finds finished data."""
    if isinstance(weight, torch.Tensor) and weight.is_cuda:
        torch.cuda.current_stream(weight.device).synchronize()


def derived_weight(weight, tag, fn):
    """A tensor computed from a parameter (negated / flipped / re-laid-out weights), cached per parameter
    version like the packed GEMM operands; host-side weight preparation, not data-path arithmetic."""
    key = (id(weight), tag)
    hit = _derived_cache.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == (weight.data_ptr(), weight._version):
        return hit[2]
    with torch.no_grad():
        out = fn(weight.detach())
        out = tuple(t.contiguous() for t in out) if isinstance(out, tuple) else out.contiguous()
    publish_prepared(weight)
    if len(_derived_cache) > 4096:
        _derived_cache.clear()
    _derived_cache[key] = (_weakref.ref(weight), (weight.data_ptr(), weight._version), out)
    return out


# ----------------------------------------------------------------------------------------
# MFMA implicit-GEMM convolution (d3d_conv_gemm_f32): weight packing, tap lists, dispatch
# ----------------------------------------------------------------------------------------
import os as _os
import weakref as _weakref

import numpy as _np

_pack_cache = {}


def _mpad(co):
    mt = (co + 15) // 16
    return 16 * (4 if mt == 3 else mt)




H16_NAMES = ("h16", "f16", "bf16")


def h16_dtype():
    """torch dtype of the library's 16-bit operand format (d3d_h16_format: "f16" by default, "bf16" in a -DD3D_H16_BF16 build):
    what channel-last "h16" volumes and packed 16-bit weight fragments are made of."""
    return torch.float16 if _lib.h16_format() == "f16" else torch.bfloat16


def _norm_precision(mode):
    """"h16" is the fast mode in whatever 16-bit format the library was built with; "f16" / "bf16" name a format and are
    accepted only when the loaded library IS that format -- asking an f16 build for bf16 must not silently run f16."""
    if mode in (None, "fp32", "h16"):
        return mode
    if mode in ("f16", "bf16"):
        if _lib.h16_format() != mode:
            raise ValueError("precision %r asked of a library whose 16-bit operand format is %r (d3d_h16_format; rebuild with "
                             "`make -C deep3d_aerial_amd/csrc H16=%s` or ask for 'h16')" % (mode, _lib.h16_format(), mode))
        return "h16"
    raise ValueError("precision must be 'fp32' or 'h16' (or the library's format by name: %r)" % _lib.h16_format())


def set_conv_precision(mode):
    """"fp32" (default; fp32 accuracy: exact fp32 MFMA or split bf16x3 operands) or "h16" (16-bit matrix-core operands in the
    library's format -- IEEE half unless built otherwise, see _lib.h16_format() -- with fp32 accumulation: BASELINE config 3's
    fast mode) for the regularisers' convolutions.  None = follow the switch table (D3D_CONV_PRECISION).
    PER THREAD (config.state is a threading.local): a forward run in a worker thread follows the switch table's
    D3D_CONV_PRECISION unless that thread calls this itself; to change the process-wide default set
    config.switches["D3D_CONV_PRECISION"]."""
    _cfg.state.conv_precision = _norm_precision(mode)


def conv_precision():
    return _cfg.state.conv_precision or _norm_precision(_cfg.get("D3D_CONV_PRECISION"))


class fp32_convs:
    """Context manager: exact fp32 convolutions inside, whatever the global precision (feature pyramids)."""

    def __enter__(self):
        self.saved = _cfg.state.conv_precision
        _cfg.state.conv_precision = "fp32"

    def __exit__(self, *exc):
        _cfg.state.conv_precision = self.saved
        return False


class h16_convs(fp32_convs):
    """Context manager: 16-bit matrix-core operands inside (BASELINE config 3's fast mode), whatever the global precision."""

    def __enter__(self):
        self.saved = _cfg.state.conv_precision
        _cfg.state.conv_precision = "h16"


def _use_mfma():
    return _cfg.get("D3D_CONV") != "direct"


def _packed(weight, transposed):
    """Packed GEMM operands of a k=3 conv weight, cached per parameter version.

    conv  [Co,Ci,(3,)3,3] -> one (wpack [T*Ci, mpad], taps int8 [T,3]) with offsets -1..1.
    convT [Ci,Co,(3,)3,3] -> one entry per output-parity class: (parity zyx, wpack, taps) with
    offsets 0/+1 (even outputs: kernel index 1 at o/2; odd: index 0 at (o+1)/2, index 2 at (o-1)/2).
    """
    # keyed by the tensor OBJECT (weak), validated by storage address and in-place version counter:
    # load_state_dict / copy_ / optimizer steps bump the version; writes through `.data` do not --
    # call clear_weight_cache() after those.
    key = (id(weight), transposed)
    hit = _pack_cache.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == (weight.data_ptr(), weight._version):
        return hit[2]
    nd = weight.dim() - 2
    with torch.no_grad():
        w = weight.detach().float()
        if nd == 2:
            w = w.unsqueeze(2)  # [.., 1, 3, 3]: z kernel of size 1
        kz_n = w.shape[2]
        if not transposed:
            Co, Ci = w.shape[0], w.shape[1]
            taps = [(kz - (kz_n // 2), ky - 1, kx - 1) for kz in range(kz_n) for ky in range(3) for kx in range(3)]
            wp = w.reshape(Co, Ci, -1).permute(2, 1, 0).reshape(-1, Co)
            pad = torch.zeros((wp.shape[0], _mpad(Co)), dtype=torch.float32, device=w.device)
            pad[:, :Co] = wp
            out = (pad.contiguous(), _np.array(taps, _np.int8).tobytes(), len(taps))
        else:
            Ci, Co = w.shape[0], w.shape[1]
            dim_opts = {0: [(1, 0)], 1: [(0, 1), (2, 0)]}  # parity -> [(kernel index, input offset)]
            out = []
            zpar = [0] if kz_n == 1 else [0, 1]
            for pz in zpar:
                for py in (0, 1):
                    for px in (0, 1):
                        zl = [(0, 0)] if kz_n == 1 else dim_opts[pz]
                        taps, cols = [], []
                        for (kz, oz) in zl:
                            for (ky, oy) in dim_opts[py]:
                                for (kx, ox) in dim_opts[px]:
                                    taps.append((oz, oy, ox))
                                    cols.append(w[:, :, kz, ky, kx])  # [Ci, Co]
                        wp = torch.stack(cols, 0).reshape(-1, Co)      # [T*Ci, Co]
                        pad = torch.zeros((wp.shape[0], _mpad(Co)), dtype=torch.float32, device=w.device)
                        pad[:, :Co] = wp
                        out.append(((pz, py, px), pad.contiguous(), _np.array(taps, _np.int8).tobytes(), len(taps)))
    publish_prepared(weight)
    if len(_pack_cache) > 4096:
        _pack_cache.clear()
    _pack_cache[key] = (_weakref.ref(weight), (weight.data_ptr(), weight._version), out)
    return out


def clear_weight_cache():
    """Drop every packed / derived weight entry (needed only after writing weights through `.data`)."""
    _pack_cache.clear()
    _derived_cache.clear()


def _gemm(x, x2, wpack, taps, ntaps, Co, scale, shift, skip, skip_after_act, act, in_dims, grid, out, istride,
          ostride, ooff):
    D, H, W = in_dims
    Dg, Hg, Wg = grid
    Do, Ho, Wo = out.shape[-3:] if out.dim() == 4 else (1,) + tuple(out.shape[-2:])
    Ci0 = x.shape[0]
    Ci1 = 0 if x2 is None else x2.shape[0]
    rc = _lib.load().d3d_conv_gemm_f32(_chk(x, "x"), Ci0, _opt(x2, "x2"), Ci1, _chk(wpack, "wpack"), wpack.shape[1],
                                       _opt(scale, "scale"), _opt(shift, "shift"), _opt(skip, "skip"),
                                       int(skip_after_act), int(act), Co, D, H, W, Dg, Hg, Wg, Do, Ho, Wo, istride,
                                       ostride, ooff[0], ooff[1], ooff[2], ntaps, taps, _chk(out, "out"), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return False
    _lib.check(rc, "d3d_conv_gemm_f32")
    return True


def conv_k3_mfma(x, weight, scale=None, shift=None, skip=None, act=0, stride=1, x2=None, skip_after_act=True):
    """k=3, pad 1 conv (2D: x [Ci,H,W]; 3D: x [Ci,D,H,W]) over cat(x, x2) on the matrix cores."""
    if _cfg.get("D3D_CONV") != "mfma_slice":
        y = conv_fold(x, weight, scale, shift, skip, act, stride, x2, skip_after_act, transposed=False)
        if y is not None:
            return y
    three_d = x.dim() == 4
    dims = tuple(x.shape[1:]) if three_d else (1,) + tuple(x.shape[1:])
    Co = weight.shape[0]
    wpack, taps, nt = _packed(weight, False)
    o = lambda n: (n - 1) // stride + 1
    od = (o(dims[0]) if three_d else 1, o(dims[1]), o(dims[2]))
    out = torch.empty((Co,) + (od if three_d else od[1:]), dtype=torch.float32, device=x.device)
    if skip is not None and skip.shape != out.shape:
        raise ValueError("skip shape %s != output shape %s" % (tuple(skip.shape), tuple(out.shape)))
    if not _gemm(x, x2, wpack, taps, nt, Co, scale, shift, skip, skip_after_act, act, dims, od, out, stride, 1,
                 (0, 0, 0)):
        return None  # shape outside the matrix-core kernels: the caller takes the direct kernel
    return out


def convtranspose_k3s2_mfma(x, weight, scale=None, shift=None, skip=None, act=0, skip_after_act=True):
    """k=3, stride 2, pad 1, output_pad 1 transposed conv as 4 (2D) / 8 (3D) parity-class launches."""
    if _cfg.get("D3D_CONV") != "mfma_slice":
        y = conv_fold(x, weight, scale, shift, skip, act, 2, None, skip_after_act, transposed=True)
        if y is not None:
            return y
    three_d = x.dim() == 4
    dims = tuple(x.shape[1:]) if three_d else (1,) + tuple(x.shape[1:])
    Co = weight.shape[1]
    od = (2 * dims[0] if three_d else 1, 2 * dims[1], 2 * dims[2])
    out = torch.empty((Co,) + (od if three_d else od[1:]), dtype=torch.float32, device=x.device)
    if skip is not None and skip.shape != out.shape:
        raise ValueError("skip shape %s != output shape %s" % (tuple(skip.shape), tuple(out.shape)))
    for (par, wpack, taps, nt) in _packed(weight, True):
        if not _gemm(x, None, wpack, taps, nt, Co, scale, shift, skip, skip_after_act, act, dims, dims, out, 1, 2, par):
            return None
    return out


# ----------------------------------------------------------------------------------------
# z-streaming folded implicit GEMM (d3d_conv_fold_f32): tap lists / packed weights per layer
# ----------------------------------------------------------------------------------------
def _dim_conv(K, stride, fold):
    """One dimension of an ordinary convolution (odd kernel size K, padding K // 2) folded over `fold`
    neighbouring outputs: (tap offsets, fold positions, k(tap, fold) -> kernel index or -1, input step, output
    step, base).  K = 0 marks the degenerate row dimension of an image (a single tap, a single position)."""
    if K == 0:
        return [0], [0], (lambda t, f: 0), 1, 1, 0
    pad = K // 2
    taps = list(range(-pad, (fold - 1) * stride + pad + 1))
    k = lambda t, f: (t - f * stride + pad) if 0 <= t - f * stride + pad < K else -1
    return taps, list(range(fold)), k, fold * stride, fold, 0


def _dim_convT(K, parities):
    """One dimension of a k=3 stride-2 pad-1 output_pad-1 transposed convolution: output 2g+p reads input
    g+o with kernel index k(o,p): p even -> (o=0: 1); p odd -> (o=0: 2, o=1: 0)."""
    if K == 0:
        return [0], [0], (lambda t, f: 0), 1, 1, 0
    table = {(0, 0): 1, (0, 1): 2, (1, 1): 0}
    k = lambda o, f: table.get((o, parities[f]), -1)
    return [0, 1], list(range(len(parities))), k, 1, 2, (parities[0] if len(parities) == 1 else 0)


def _fold_pack(wk, dims, Co, ksizes):
    """wk [Co,Ci,K0*K1*K2] (kernel sizes ksizes, in the streaming/row/column order of `dims`) ->
    (wpack [T,Ci,mpad], taps bytes (sorted by the first dimension), T, M, mpad, [c,s,b,f per dim])."""
    (t0, f0, k0, c0, s0, b0), (t1, f1, k1, c1, s1, b1), (t2, f2, k2, c2, s2, b2) = dims
    K1, K2 = ksizes[1], ksizes[2]
    F = len(f0) * len(f1) * len(f2)
    M = Co * F
    if M > 64:
        raise ValueError("fold %dx%dx%d of %d channels exceeds 64 GEMM rows" % (len(f0), len(f1), len(f2), Co))
    mpad = 16 if M <= 16 else (32 if M <= 32 else 64)
    nk = wk.shape[2]
    taps, kidx = [], []
    for o0 in t0:
        for o1 in t1:
            for o2 in t2:
                row = []
                for a in f0:
                    for b in f1:
                        for c in f2:
                            i, jj, l = k0(o0, a), k1(o1, b), k2(o2, c)
                            row.append(nk if min(i, jj, l) < 0 else (i * K1 + jj) * K2 + l)
                if any(r != nk for r in row):
                    taps.append((o0, o1, o2))
                    kidx.append(row)
    T = len(taps)
    Ci = wk.shape[1]
    wz = torch.cat([wk, torch.zeros((Co, Ci, 1), dtype=wk.dtype, device=wk.device)], 2)
    idx = torch.tensor(kidx, dtype=torch.long, device=wk.device)        # [T, F]
    a = wz[:, :, idx]                                                    # [Co, Ci, T, F]
    a = a.permute(2, 1, 3, 0).reshape(T, Ci, M)                          # row m = fold*Co + co
    wpack = torch.zeros((T, Ci, mpad), dtype=torch.float32, device=wk.device)
    wpack[:, :, :M] = a
    tail = [c0, c1, c2, s0, s1, s2, b0, b1, b2, len(f0), len(f1), len(f2)]
    return wpack.contiguous(), _np.array(taps, _np.int8).tobytes(), T, M, mpad, tail


def _conv_fold_choice(Co, Ci, three_d, stride, K=3):
    """Fold (f_y, f_x) that fills the 16 GEMM rows of a narrow layer.  Limits: the kernel's 128 taps, and resident
    weights (ntaps * Ci * 16 floats) small enough that two workgroups still share a CU's LDS -- a wide-C_in layer
    is faster unfolded at twice the occupancy (stage-1 conv0 32->8: 9.4 ms folded, 5.3 ms unfolded)."""
    budget = 48 * 1024
    ntaps = lambda f: (K if three_d else 1) * ((f[0] - 1) * stride + K) * ((f[1] - 1) * stride + K)
    for f in [(4, 4), (2, 4), (2, 2), (1, 2)]:
        # (the kernel's column step f_x * stride must be 1, 2 or 4)
        if Co * f[0] * f[1] <= 16 and f[1] * stride <= 4 and ntaps(f) <= 128 and ntaps(f) * Ci * 64 <= budget:
            return f
    return (1, 1)


def _packed_fold(weight, transposed, stride):
    """List of launches [(wpack, taps, T, M, mpad, geom tail)] for one layer, cached like _packed.

    Dimension order of the kernel is (streamed, row, column).  A volume [C,D,H,W] maps (z, y, x) onto it; an
    image [C,H,W] is handed over as [C, H, 1, W] -- its rows are the streamed planes, so every input row is
    staged once and the kernel's z machinery (open accumulator sets, z fold) serves the image's y axis."""
    key = (id(weight), "fold", transposed, stride)
    hit = _pack_cache.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == (weight.data_ptr(), weight._version):
        return hit[2]
    with torch.no_grad():
        w = weight.detach().float()
        three_d = w.dim() == 5
        if transposed:
            w = w.transpose(0, 1)
        Co, Ci = w.shape[0], w.shape[1]
        wk = w.reshape(Co, Ci, -1).contiguous()
        K = w.shape[-1]  # cubic / square kernels, odd size, padding K // 2 (1, 3 and 5 occur in the reference)
        if K % 2 == 0 or any(d != K for d in w.shape[2:]) or (transposed and K != 3):
            raise ValueError("unsupported kernel shape %s" % (tuple(w.shape[2:]),))
        ks = (K, K, K) if three_d else (K, 1, K)
        launches = []
        if not transposed:
            fy, fx = _conv_fold_choice(Co, Ci, three_d, stride, K)
            if three_d:
                dims = (_dim_conv(K, stride, 1), _dim_conv(K, stride, fy), _dim_conv(K, stride, fx))
            else:
                dims = (_dim_conv(K, stride, fy), _dim_conv(0, 1, 1), _dim_conv(K, stride, fx))
            launches.append(_fold_pack(wk, dims, Co, ks))
        else:
            # all output parities as GEMM rows while they fit 64 rows; otherwise one launch per parity of the
            # leading dimensions
            ks0 = [k if k == 3 else 0 for k in ks]
            sets = [[[0, 1]] if k == 3 else [None] for k in ks]
            rows = lambda: Co * int(_np.prod([len(ss[0]) if ss[0] else 1 for ss in sets]))
            for d in range(3):
                if rows() > 64 and ks[d] == 3:
                    sets[d] = [[0], [1]]
            for p0 in sets[0]:
                for p1 in sets[1]:
                    for p2 in sets[2]:
                        dims = tuple(_dim_convT(ks0[d], pp) for d, pp in enumerate((p0, p1, p2)))
                        launches.append(_fold_pack(wk, dims, Co, ks))
    publish_prepared(weight)
    if len(_pack_cache) > 4096:
        _pack_cache.clear()
    _pack_cache[key] = (_weakref.ref(weight), (weight.data_ptr(), weight._version), launches)
    return launches


def conv_fold(x, weight, scale=None, shift=None, skip=None, act=0, stride=1, x2=None, skip_after_act=True,
              transposed=False, aux1=None, ep_split=0):
    """k=3 convolution (2D [C,H,W] or 3D [C,D,H,W]; ordinary stride 1|2 over cat(x,x2), or transposed
    stride 2) through the z-streaming folded GEMM.  Returns None when the layer's resident weights do not fit
    LDS (D3D_ERR_UNSUPPORTED): the caller then takes the per-slice MFMA path."""
    three_d = x.dim() == 4
    # kernel dimension order (streamed, row, column): a volume is (D,H,W), an image (H,1,W) -- see _packed_fold
    D, H, W = tuple(x.shape[1:]) if three_d else (x.shape[1], 1, x.shape[2])
    Co = weight.shape[1] if transposed else weight.shape[0]
    if transposed:
        od = (2 * D, 2 * H if three_d else 1, 2 * W)
    else:
        o = lambda n: (n - 1) // stride + 1
        od = (o(D), o(H) if three_d else 1, o(W))
    out = torch.empty((Co,) + (od if three_d else (od[0], od[2])), dtype=torch.float32, device=x.device)
    if skip is not None and skip.shape != out.shape and act < 2:
        raise ValueError("skip shape %s != output shape %s" % (tuple(skip.shape), tuple(out.shape)))
    lib = _lib.load()
    Ci0 = x.shape[0]
    Ci1 = 0 if x2 is None else x2.shape[0]
    for (wpack, taps, T, M, mpad, tail) in _packed_fold(weight, transposed, stride):
        if wpack.shape[1] != Ci0 + Ci1:
            raise ValueError("weight has %d input channels, inputs have %d" % (wpack.shape[1], Ci0 + Ci1))
        cz, cy, cx, sz, sy, sx, bz, by, bx, fz, fy, fx = tail
        if transposed:
            G = (D, H, W)
        else:
            G = tuple((od[i] + (fz, fy, fx)[i] - 1) // (fz, fy, fx)[i] for i in range(3))
        geom = (ctypes.c_int * 15)(G[0], G[1], G[2], cz, cy, cx, sz, sy, sx, bz, by, bx, fz, fy, fx)
        fold = lib.d3d_conv_fold_h16 if conv_precision() == "h16" else lib.d3d_conv_fold_f32
        rc = fold(_chk(x, "x"), Ci0, _opt(x2, "x2"), Ci1, _chk(wpack, "wpack"), mpad, M,
                                   _opt(scale, "scale"), _opt(shift, "shift"), _opt(skip, "skip"),
                                   int(skip_after_act), int(act), _opt(aux1, "aux1"), int(ep_split), Co, D, H, W,
                                   od[0], od[1], od[2], geom, T, taps, _chk(out, "out"), _stream())
        if rc == _lib.ERR_UNSUPPORTED and fold is lib.d3d_conv_fold_h16:
            # shape outside the bf16 kernels (e.g. image width not a multiple of 4): the exact fp32 kernel instead
            rc = lib.d3d_conv_fold_f32(_chk(x, "x"), Ci0, _opt(x2, "x2"), Ci1, _chk(wpack, "wpack"), mpad, M,
                                       _opt(scale, "scale"), _opt(shift, "shift"), _opt(skip, "skip"),
                                       int(skip_after_act), int(act), _opt(aux1, "aux1"), int(ep_split), Co, D, H, W,
                                       od[0], od[1], od[2], geom, T, taps, _chk(out, "out"), _stream())
        if rc == _lib.ERR_UNSUPPORTED:
            return None
        _lib.check(rc, "d3d_conv_fold")
    return out


def conv2d_same(x, weight, scale=None, shift=None, skip=None, act=0, stride=1):
    """Conv2d with an odd square kernel (1, 3, 5, ...) and padding K // 2, stride 1 | 2, on the matrix-core stream
    kernel (feature pyramids: module.py:657-679 5x5 stride-2 and 1x1 layers).  Returns None when the layer does
    not fit that kernel (the caller then uses MIOpen)."""
    Co, Ci = weight.shape[0], weight.shape[1]
    if x.shape[0] != Ci or Co > 64 or not _use_mfma():
        return None
    if weight.shape[2] == 1 and stride == 1 and conv_precision() != "h16" and not _cfg.off("conv2d_k1"):
        y = conv2d_k1(x, weight, scale, shift, skip, act)
        if y is not None:
            return y
    if weight.shape[2] == 5 and stride == 2 and conv_precision() != "h16":
        y = conv2d_k5s2_zs(x, weight, scale, shift, skip, act)
        if y is not None:
            return y
    return conv_fold(x, weight, scale, shift, skip, act, stride, None, True, transposed=False)


def _pack_k1(w):
    """[Co,Ci,1,1] -> [ceil(Co / 8)][Ci][8]: blocks of 8 output channels, zero-padded (d3d_conv2d_k1_f32)."""
    Co, Ci = w.shape[0], w.shape[1]
    nb = (Co + 7) // 8
    wp = w.new_zeros((nb * 8, Ci))
    wp[:Co] = w.reshape(Co, Ci)
    return wp.reshape(nb, 8, Ci).permute(0, 2, 1).contiguous()


def conv2d_k1(x, weight, scale=None, shift=None, skip=None, act=0):
    """Conv2d(k 1) in exact fp32 as a streaming kernel (d3d_conv2d_k1_f32: the 1 x 1 output layers of the feature pyramids);
    act(scale * conv + shift) + skip.  None for shapes it does not take."""
    Ci, H, W = x.shape
    Co = weight.shape[0]
    if Ci not in (8, 16, 32) or (H * W) % 4 or act not in (0, 1) or tuple(weight.shape) != (Co, Ci, 1, 1):
        return None
    wp = derived_weight(weight, "k1f32", _pack_k1)
    out = torch.empty((Co, H, W), dtype=torch.float32, device=x.device)
    if skip is not None and skip.shape != out.shape:
        raise ValueError("skip shape mismatch")
    rc = _lib.load().d3d_conv2d_k1_f32(_chk(x, "x", 3), _chk(wp, "wpacked"), _opt(scale, "scale"), _opt(shift, "shift"), _opt(skip, "skip"),
                                       int(act), Ci, Co, H, W, _chk(out, "out"), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "d3d_conv2d_k1_f32")
    dispatch_counts["conv2d_k1"] += 1
    return out


def conv2d_k5s2_zs(x, weight, scale=None, shift=None, skip=None, act=0, skip_after_act=True):
    """Conv2d(k 5, stride 2, pad 2) on the stride-2 tile kernel with split operands (fp32 accuracy; the downsampling layers of
    the feature trunks, d3d_conv2d_k5s2_zs_bf16x3); None for shapes it does not take."""
    Ci, H, W = x.shape
    Co = weight.shape[0]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if (Ci, Co <= 16) != (8, True) and (Ci, Co <= 32) != (16, True) or Wo % 4 or act not in (0, 1) or H * W < 128 * 128 \
            or tuple(weight.shape) != (Co, Ci, 5, 5) or _z2_fp32_entry() != "x3" or _cfg.off("conv2d_zs") \
            or _cfg.off("conv2d_k5"):
        return None
    wp = derived_weight(weight, "z2k5bf16x3", _pack_z2_bf16x3)
    out = torch.empty((Co, Ho, Wo), dtype=torch.float32, device=x.device)
    if skip is not None and skip.shape != out.shape:
        raise ValueError("skip shape mismatch")
    rc = _lib.load().d3d_conv2d_k5s2_zs_bf16x3(_chk(x, "x", 3), ctypes.c_void_p(wp.data_ptr()), _opt(scale, "scale"), _opt(shift, "shift"),
                                               _opt(skip, "skip"), int(act), int(bool(skip_after_act)), Ci, Co, H, W, _chk(out, "out"),
                                               _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "d3d_conv2d_k5s2_zs_bf16x3")
    return out


def gru_cell_conv_fused(cost, h, w_pre, w_gates, b_gates, w_cand, b_cand, stride=1, out=None):
    """relu(conv3x3(cost, stride)) followed by the conv-GRU cell on it, ONE launch (d3d_gru_cell_fused_h16, csrc/gru_fused.hip;
    adamvs.py:409-412: conv1 + conv_gru1 at stride 1, conv2 + conv_gru2 at stride 2).  bf16 mode only (the operands are bf16, the
    state stays fp32): bit-identical to conv2d_zs + gru_cell_fused.  Returns the new state, or None for shapes / modes the
    kernel does not take (the caller then runs the separate launches)."""
    if conv_precision() != "h16" or not _use_mfma() or _cfg.off("gru_fused") or cost.dim() not in (3, 4):
        return None
    if cost.dim() == 4:   # a CL8 plane [C/8, H, W, 8] of 16-bit cells (weighted_corr_cl8): the stride-1 cell's own entry point
        G8, HI, WI, _ = cost.shape
        HID, H, W = h.shape
        CP = 8 * G8
        if cost.dtype != h16_dtype() or cost.shape[3] != 8 or not cost.is_contiguous() or stride != 1 or (HI, WI) != (H, W) \
                or CP not in (8, 16, 32) or HID != 8 or b_gates is None or b_cand is None:
            return None
        w1 = derived_weight(w_pre, "z2bf16", _pack_z2_bf16)
        wg = derived_weight(w_gates, "z2bf16", _pack_z2_bf16)
        wc = derived_weight(w_cand, "z2bf16", _pack_z2_bf16)
        if out is None:
            out = torch.empty_like(h)
        elif out.shape != h.shape or out.dtype != h.dtype or out.data_ptr() == h.data_ptr():
            raise ValueError("out must be a separate tensor of the state's shape")
        rc = _lib.load().d3d_gru_cell_fused_cl8_h16(ctypes.c_void_p(cost.data_ptr()), CP, _chk(h, "h", 3), HID, H, W,
                                                    ctypes.c_void_p(w1.data_ptr()), ctypes.c_void_p(wg.data_ptr()), _chk(b_gates, "b_gates"),
                                                    ctypes.c_void_p(wc.data_ptr()), _chk(b_cand, "b_cand"), _chk(out, "out"), _stream())
        if rc == _lib.ERR_UNSUPPORTED:
            return None
        _lib.check(rc, "d3d_gru_cell_fused_cl8_h16")
        dispatch_counts["gru_cell_fused"] += 1
        return out
    CP, HI, WI = cost.shape
    HID, H, W = h.shape
    if stride == 1:
        ok = CP in (8, 16, 32) and HID == 8 and (HI, WI) == (H, W)
    else:
        ok = stride == 2 and CP == 8 and HID == 16 and (H, W) == ((HI - 1) // 2 + 1, (WI - 1) // 2 + 1)
    if not ok or b_gates is None or b_cand is None:
        return None
    if tuple(w_pre.shape) != (HID, CP, 3, 3) or tuple(w_gates.shape) != (2 * HID, 2 * HID, 3, 3) or tuple(w_cand.shape) != (HID, 2 * HID, 3, 3):
        raise ValueError("conv-GRU cell weights do not match C = %d, hidden = %d" % (CP, HID))
    w1 = derived_weight(w_pre, "z2bf16", _pack_z2_bf16)
    wg = derived_weight(w_gates, "z2bf16", _pack_z2_bf16)
    wc = derived_weight(w_cand, "z2bf16", _pack_z2_bf16)
    if out is None:
        out = torch.empty_like(h)
    elif out.shape != h.shape or out.dtype != h.dtype or out.data_ptr() == h.data_ptr():
        raise ValueError("out must be a separate tensor of the state's shape")
    rc = _lib.load().d3d_gru_cell_fused_h16(_chk(cost, "cost", 3), CP, HI, WI, int(stride), _chk(h, "h", 3), HID, H, W,
                                             ctypes.c_void_p(w1.data_ptr()), ctypes.c_void_p(wg.data_ptr()), _chk(b_gates, "b_gates"),
                                             ctypes.c_void_p(wc.data_ptr()), _chk(b_cand, "b_cand"), _chk(out, "out"), _stream())
    if rc == _lib.ERR_UNSUPPORTED:
        return None
    _lib.check(rc, "d3d_gru_cell_fused_h16")
    dispatch_counts["gru_cell_fused"] += 1
    return out


def gru_cell_fused(x, h, w_gates, b_gates, w_cand, b_cand):
    """ConvGRUCell.forward (module.py:24-51) as two convolutions with fused epilogues: gates conv -> [r*h | u], then
    the candidate conv over cat(x, r*h) -> h' = u*h + (1-u)*tanh(c).  Returns None when the layer does not fit the
    image stream kernel (the caller then runs the four-kernel form)."""
    Hc = h.shape[0]
    if not _use_mfma() or _cfg.get("D3D_CONV") == "mfma_slice" or x.dim() != 3 or 2 * Hc > 64:
        return None
    if x.shape[1] * x.shape[2] >= 128 * 128:
        # both convolutions on the tile kernel (v_mfma_f32_16x16x32_bf16 in bf16 mode, exact v_mfma_f32_16x16x4_f32
        # otherwise; 16-byte epilogue accesses)
        g = conv2d_zs(x, w_gates, None, b_gates, h, 2, x2=h, ep_split=Hc)
        if g is not None:
            hn = conv2d_zs(x, w_cand, None, b_cand, h, 3, x2=g[:Hc], aux1=g[Hc:])
            if hn is not None:
                return hn
    # both convolutions of a small cell run on the vector-unit kernel (AdaMVS view, same device: 87.9 ms, with the
    # gates on the matrix cores 89.3 ms)
    g = conv2d_stream(x, w_gates, None, b_gates, h, 2, x2=h, ep_split=Hc)
    if g is None:
        g = conv_fold(x, w_gates, None, b_gates, h, act=2, stride=1, x2=h, ep_split=Hc)
    if g is None:
        g = conv2d_stream(x, w_gates, None, b_gates, h, 2, x2=h, ep_split=Hc)
    if g is None:
        return None
    rh, u = g[:Hc], g[Hc:]
    hn = conv2d_stream(x, w_cand, None, b_cand, h, 3, x2=rh, aux1=u)
    if hn is None:
        hn = conv_fold(x, w_cand, None, b_cand, h, act=3, stride=1, x2=rh, aux1=u)
    return hn
