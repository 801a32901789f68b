"""Tensor-level front end of the C-ABI kernels (one function per entry point).

PyTorch is plumbing here: it owns device memory and the current HIP stream; all arithmetic
happens in libdeep3d_planesweep.so.  Tensors must be fp32, contiguous and on the GPU --
anything else raises (no CPU path exists).  Shapes are unbatched, as in the header.
"""
import ctypes

import torch

from . import _lib

PER_PLANE, PER_PIXEL = 0, 1


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


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


def _opt(t, name):
    return None if t is None else _chk(t, name)


def _depth(depth, h, w):
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
    dp, mode, D = _depth(depth, h, w)
    if out is None:
        out = torch.empty((C, D, h, w), dtype=torch.float32, device=src.device)
    rc = _lib.load().d3d_homo_warp(_chk(src, "src", 3), _chk(proj34, "proj34"), dp, mode, C, D, h, w,
                                   _chk(out, "out", 4), _stream())
    _lib.check(rc, "d3d_homo_warp")
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


def variance_volume(feats, proj34, depth, out=None):
    """feats: list of V tensors [C,h,w] (feats[0] = reference); -> [C,D,h,w] (cas_mvsnet.py:45-60)."""
    C, h, w = _check_feats(feats, proj34)
    dp, mode, D = _depth(depth, h, w)
    if out is None:
        out = torch.empty((C, D, h, w), dtype=torch.float32, device=feats[0].device)
    arr = _ptr_array(feats, "feats")
    rc = _lib.load().d3d_variance_volume(arr, _chk(proj34, "proj34"), dp, mode, len(feats), C, D, h, w,
                                         _chk(out, "out", 4), _stream())
    _lib.check(rc, "d3d_variance_volume")
    return out


def weighted_corr(feats, proj34, weights, depth, out=None):
    """adamvs.py:492-509. weights [V-1,h,w] -> [C,D,h,w]."""
    C, h, w = _check_feats(feats, proj34)
    if tuple(weights.shape) != (len(feats) - 1, h, w):
        raise ValueError("weights must be [%d,%d,%d]" % (len(feats) - 1, h, w))
    dp, mode, D = _depth(depth, h, w)
    if out is None:
        out = torch.empty((C, D, h, w), dtype=torch.float32, device=feats[0].device)
    arr = _ptr_array(feats, "feats")
    rc = _lib.load().d3d_weighted_corr(arr, _chk(proj34, "proj34"), _chk(weights, "weights", 3), dp, mode,
                                       len(feats), C, D, h, w, _chk(out, "out", 4), _stream())
    _lib.check(rc, "d3d_weighted_corr")
    return out


def pair_corr_mean(ref, src, proj34, depth, out=None):
    """adamvs.py:469-474. -> [D,h,w]."""
    C, h, w = _check_feats([ref, src], proj34)
    dp, mode, D = _depth(depth, h, w)
    if out is None:
        out = torch.empty((D, h, w), dtype=torch.float32, device=ref.device)
    rc = _lib.load().d3d_pair_corr_mean(_chk(ref, "ref", 3), _chk(src, "src", 3), _chk(proj34, "proj34"), dp, mode,
                                        C, D, h, w, _chk(out, "out", 3), _stream())
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


def pair_softmax_max(score, depth):
    """score [D,h,w] -> (view_weight [h,w], pair_depth [h,w]) (adamvs.py:478-486)."""
    D, h, w = score.shape
    dp, mode, D2 = _depth(depth, h, w)
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


def resize_bilinear(x, H, W):
    """[n,h,w] -> [n,H,W], F.interpolate(bilinear, align_corners=False) semantics."""
    n, h, w = x.shape
    out = torch.empty((n, H, W), dtype=torch.float32, device=x.device)
    rc = _lib.load().d3d_resize_bilinear(_chk(x, "x", 3), n, h, w, H, W, _chk(out, "out"), _stream())
    _lib.check(rc, "d3d_resize_bilinear")
    return out


def conv3d_k3(x, weight, scale=None, shift=None, skip=None, relu=True, stride=1):
    """x [Ci,D,H,W], weight [Co,Ci,3,3,3] -> [Co,Do,Ho,Wo] with folded-BN affine, ReLU, skip (after ReLU)."""
    Ci, D, H, W = x.shape
    Co = weight.shape[0]
    if tuple(weight.shape) != (Co, Ci, 3, 3, 3):
        raise ValueError("weight must be [Co,%d,3,3,3] (got %s)" % (Ci, tuple(weight.shape)))
    o = lambda n: (n - 1) // stride + 1
    out = torch.empty((Co, o(D), o(H), o(W)), dtype=torch.float32, device=x.device)
    if skip is not None and skip.shape != out.shape:
        raise ValueError("skip shape %s != output shape %s" % (tuple(skip.shape), tuple(out.shape)))
    rc = _lib.load().d3d_conv3d_k3(_chk(x, "x", 4), _chk(weight, "weight"), _opt(scale, "scale"),
                                   _opt(shift, "shift"), _opt(skip, "skip"), int(relu), Ci, Co, D, H, W, stride,
                                   _chk(out, "out"), _stream())
    _lib.check(rc, "d3d_conv3d_k3")
    return out


def convtranspose3d_k3s2(x, weight, scale=None, shift=None, skip=None, relu=True):
    """x [Ci,D,H,W], weight [Ci,Co,3,3,3] -> [Co,2D,2H,2W]."""
    Ci, D, H, W = x.shape
    Co = weight.shape[1]
    if tuple(weight.shape) != (Ci, Co, 3, 3, 3):
        raise ValueError("weight must be [%d,Co,3,3,3] (got %s)" % (Ci, tuple(weight.shape)))
    out = torch.empty((Co, 2 * D, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    if skip is not None and skip.shape != out.shape:
        raise ValueError("skip shape %s != output shape %s" % (tuple(skip.shape), tuple(out.shape)))
    rc = _lib.load().d3d_convtranspose3d_k3s2(_chk(x, "x", 4), _chk(weight, "weight"), _opt(scale, "scale"),
                                              _opt(shift, "shift"), _opt(skip, "skip"), int(relu), Ci, Co, D, H, W,
                                              _chk(out, "out"), _stream())
    _lib.check(rc, "d3d_convtranspose3d_k3s2")
    return out


def conv2d_k3(x, weight, scale=None, shift=None, skip=None, act=0, stride=1, x2=None):
    """3x3 conv over cat(x, x2) channels. x [Ci0,H,W], x2 [Ci1,H,W]|None, weight [Co,Ci0+Ci1,3,3]."""
    Ci0, H, W = x.shape
    Ci1 = 0 if x2 is None else x2.shape[0]
    Co = weight.shape[0]
    if tuple(weight.shape) != (Co, Ci0 + Ci1, 3, 3):
        raise ValueError("weight must be [Co,%d,3,3] (got %s)" % (Ci0 + Ci1, tuple(weight.shape)))
    if x2 is not None and tuple(x2.shape[1:]) != (H, W):
        raise ValueError("x2 spatial size mismatch")
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
