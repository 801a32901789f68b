"""numpy front-end of libplanesweep_oracle.so (see planesweep_oracle.c for the citations).

TEST INFRASTRUCTURE ONLY -- never imported by deep3d_aerial_amd.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# D3D_ORACLE_SO: use a pre-built variant instead (e.g. the -fsanitize=address,undefined build of `make asan`)
_SO = os.environ.get("D3D_ORACLE_SO") or os.path.join(_HERE, "libplanesweep_oracle.so")

_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """Compile the oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "planesweep_oracle.c")
    if os.environ.get("D3D_ORACLE_SO"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libplanesweep_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(_f32p)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads():
    return int(lib().d3d_oracle_num_threads())


def set_num_threads(n):
    lib().d3d_oracle_set_num_threads(int(n))


def compose_proj(src44, ref44):
    """[4,4],[4,4] -> [3,4] = (src @ inv(ref))[:3] as [rot|trans]."""
    src44, ref44 = _c(src44), _c(ref44)
    out = np.empty((3, 4), np.float32)
    rc = lib().d3d_oracle_compose_proj(_p(src44), _p(ref44), _p(out))
    if rc != 0:
        raise np.linalg.LinAlgError("singular ref_proj")
    return out


def _depth_args(depth, h, w):
    depth = _c(depth)
    if depth.ndim == 1:
        return depth, 0, depth.shape[0]
    assert depth.shape[1:] == (h, w), (depth.shape, h, w)
    return depth, 1, depth.shape[0]


def homo_warp(src, proj34, depth):
    src, proj34 = _c(src), _c(proj34)
    C, h, w = src.shape
    depth, is_map, D = _depth_args(depth, h, w)
    out = np.empty((C, D, h, w), np.float32)
    lib().d3d_oracle_homo_warp(_p(src), _p(proj34), _p(depth), is_map, C, D, h, w, _p(out))
    return out


def homo_warp_double(src, src_proj44, ref_proj44, depth):
    """module.py:560-601: projection matrices in float64 [4,4]."""
    import ctypes

    src = _c(src)
    C, h, w = src.shape
    depth, is_map, D = _depth_args(depth, h, w)
    out = np.empty((C, D, h, w), np.float32)
    r = np.ascontiguousarray(ref_proj44, np.float64)
    s_ = np.ascontiguousarray(src_proj44, np.float64)
    pd = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    rc = lib().d3d_oracle_homo_warp_f64(_p(src), pd(r), pd(s_), _p(depth), is_map, C, D, h, w, _p(out))
    if rc != 0:
        raise ValueError("singular reference projection")
    return out


def variance_volume(ref, srcs, projs, depth):
    """ref [C,h,w]; srcs [V-1,C,h,w]; projs [V-1,3,4]; depth [D] or [D,h,w] -> [C,D,h,w]."""
    ref, srcs, projs = _c(ref), _c(srcs), _c(projs)
    C, h, w = ref.shape
    V = srcs.shape[0] + 1
    depth, is_map, D = _depth_args(depth, h, w)
    out = np.empty((C, D, h, w), np.float32)
    lib().d3d_oracle_variance_volume(_p(ref), _p(srcs), _p(projs), _p(depth), is_map, V, C, D, h, w, _p(out))
    return out


def pair_corr_mean(ref, src, proj34, depth):
    ref, src, proj34 = _c(ref), _c(src), _c(proj34)
    C, h, w = ref.shape
    depth, is_map, D = _depth_args(depth, h, w)
    out = np.empty((D, h, w), np.float32)
    lib().d3d_oracle_pair_corr_mean(_p(ref), _p(src), _p(proj34), _p(depth), is_map, C, D, h, w, _p(out))
    return out


def weighted_corr(ref, srcs, projs, weights, depth):
    ref, srcs, projs, weights = _c(ref), _c(srcs), _c(projs), _c(weights)
    C, h, w = ref.shape
    V = srcs.shape[0] + 1
    assert weights.shape == (V - 1, h, w)
    depth, is_map, D = _depth_args(depth, h, w)
    out = np.empty((C, D, h, w), np.float32)
    lib().d3d_oracle_weighted_corr(_p(ref), _p(srcs), _p(projs), _p(weights), _p(depth), is_map, V, C, D, h, w,
                                   _p(out))
    return out


def softargmin_conf4(cost, depth):
    cost = _c(cost)
    D, h, w = cost.shape
    depth, is_map, D2 = _depth_args(depth, h, w)
    assert D2 == D
    dep = np.empty((h, w), np.float32)
    conf = np.empty((h, w), np.float32)
    lib().d3d_oracle_softargmin_conf4(_p(cost), _p(depth), is_map, D, h, w, _p(dep), _p(conf))
    return dep, conf


def softargmin_conf4_var(cost, depth, lamb):
    """ucsnet.py:137-151 (compute_depth): softmax over D, expected depth, 4-plane confidence at the expected index, and
    exp_variance = lamb * sqrt(sum_d p_d (depth_d - depth)^2).  numpy restatement (fp32 arithmetic, in the reference's order)."""
    cost = _c(cost)
    D, h, w = cost.shape
    depth, is_map, D2 = _depth_args(depth, h, w)
    assert D2 == D
    dv = depth.reshape(D, h, w) if is_map else np.broadcast_to(depth.reshape(D, 1, 1), (D, h, w))
    dep, conf = softargmin_conf4(cost, depth)
    e = np.exp(cost - cost.max(0, keepdims=True), dtype=np.float32)
    prob = (e / e.sum(0, keepdims=True, dtype=np.float32)).astype(np.float32)
    samp = ((dv - dep[None]) ** 2).astype(np.float32)
    var = np.float32(lamb) * np.sqrt((samp * prob).sum(0, dtype=np.float32), dtype=np.float32)
    return dep, conf, var.astype(np.float32)


def uncertainty_aware_samples(cur_depth, exp_var, ndepth):
    """ucsnet.py:30-53.  cur_depth [2+] (first stage): min + arange(ndepth) * (max - min) / (ndepth - 1) -> [ndepth];
    cur_depth, exp_var [h,w]: low + step * i + 1e-12, low = cur - var, step = ((cur + var) - low) / (ndepth - 1)."""
    cur = np.asarray(cur_depth, np.float32)
    if cur.ndim == 1:
        interval = np.float32((cur[-1] - cur[0]) / np.float32(ndepth - 1))
        return (cur[0] + np.arange(ndepth, dtype=np.float32) * interval).astype(np.float32)
    var = np.asarray(exp_var, np.float32)
    low, high = cur - var, cur + var
    step = ((high - low) / np.float32(float(ndepth) - 1)).astype(np.float32)
    return np.stack([(low + step * np.float32(i) + np.float32(1e-12)).astype(np.float32) for i in range(int(ndepth))])


def resize_bilinear(x, H, W):
    x = _c(x)
    h, w = x.shape
    out = np.empty((H, W), np.float32)
    lib().d3d_oracle_resize_bilinear(_p(x), h, w, H, W, _p(out))
    return out


def online_regress(regs, dplanes):
    """regs, dplanes: [D,H,W] (dplanes already at output resolution) -> depth, conf [H,W]."""
    regs, dplanes = _c(regs), _c(dplanes)
    D, H, W = regs.shape
    n = H * W
    mx = np.zeros(n, np.float32)
    sd = np.zeros(n, np.float32)
    sp = np.zeros(n, np.float32)
    L = lib()
    L.d3d_oracle_online_regress_update.argtypes = [_f32p, _f32p, ctypes.c_long, _f32p, _f32p, _f32p]
    for d in range(D):
        L.d3d_oracle_online_regress_update(_p(np.ascontiguousarray(regs[d])), _p(np.ascontiguousarray(dplanes[d])),
                                           n, _p(mx), _p(sd), _p(sp))
    dep = np.empty(n, np.float32)
    conf = np.empty(n, np.float32)
    L.d3d_oracle_online_regress_finalize.argtypes = [_f32p, _f32p, _f32p, ctypes.c_long, _f32p, _f32p]
    L.d3d_oracle_online_regress_finalize(_p(mx), _p(sd), _p(sp), n, _p(dep), _p(conf))
    return dep.reshape(H, W), conf.reshape(H, W)


def depth_range_samples(cur_depth, D, interval, h, w):
    cur_depth = _c(cur_depth)
    mode = 0 if cur_depth.ndim == 1 else 1
    out = np.empty((D, h, w), np.float32)
    L = lib()
    L.d3d_oracle_depth_range_samples.argtypes = [_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                                 ctypes.c_int, _f32p]
    L.d3d_oracle_depth_range_samples(_p(cur_depth), mode, D, float(interval), h, w, _p(out))
    return out


# ---- convolution family (naive) -------------------------------------------------------

def conv3d_k3(x, wt, bias=None, stride=1):
    x, wt = _c(x), _c(wt)
    Ci, D, H, W = x.shape
    Co = wt.shape[0]
    o = lambda n: (n + 2 - 3) // stride + 1
    out = np.empty((Co, o(D), o(H), o(W)), np.float32)
    b = _c(bias) if bias is not None else None
    lib().d3d_oracle_conv3d_k3(_p(x), _p(wt), _p(b), Ci, Co, D, H, W, stride, _p(out))
    return out


def convtranspose3d_k3s2(x, wt, bias=None):
    x, wt = _c(x), _c(wt)
    Ci, D, H, W = x.shape
    Co = wt.shape[1]
    out = np.empty((Co, 2 * D, 2 * H, 2 * W), np.float32)
    b = _c(bias) if bias is not None else None
    lib().d3d_oracle_convtranspose3d_k3s2(_p(x), _p(wt), _p(b), Ci, Co, D, H, W, _p(out))
    return out


def conv2d_k3(x, wt, bias=None, stride=1):
    x, wt = _c(x), _c(wt)
    Ci, H, W = x.shape
    Co = wt.shape[0]
    o = lambda n: (n + 2 - 3) // stride + 1
    out = np.empty((Co, o(H), o(W)), np.float32)
    b = _c(bias) if bias is not None else None
    lib().d3d_oracle_conv2d_k3(_p(x), _p(wt), _p(b), Ci, Co, H, W, stride, _p(out))
    return out


def convtranspose2d_k3s2(x, wt, bias=None):
    x, wt = _c(x), _c(wt)
    Ci, H, W = x.shape
    Co = wt.shape[1]
    out = np.empty((Co, 2 * H, 2 * W), np.float32)
    b = _c(bias) if bias is not None else None
    lib().d3d_oracle_convtranspose2d_k3s2(_p(x), _p(wt), _p(b), Ci, Co, H, W, _p(out))
    return out


def bn_relu_add(x, gamma, beta, mean, var, eps=1e-5, relu=True, skip=None):
    """Eval-mode BN + optional ReLU + optional skip add (after the ReLU). Returns a new array."""
    x = _c(x).copy()
    C = x.shape[0]
    n = x.size // C
    L = lib()
    L.d3d_oracle_bn_relu_add.argtypes = [_f32p, ctypes.c_int, ctypes.c_long, _f32p, _f32p, _f32p, _f32p,
                                         ctypes.c_float, ctypes.c_int, _f32p]
    s = _c(skip) if skip is not None else None
    L.d3d_oracle_bn_relu_add(_p(x), C, n, _p(_c(gamma)), _p(_c(beta)), _p(_c(mean)), _p(_c(var)), float(eps),
                             int(relu), _p(s))
    return x


def sigmoid(x):
    x = _c(x).copy()
    L = lib()
    L.d3d_oracle_sigmoid.argtypes = [_f32p, ctypes.c_long]
    L.d3d_oracle_sigmoid(_p(x), x.size)
    return x


def gru_update(u, h, convc):
    u, h, convc = _c(u), _c(h), _c(convc)
    out = np.empty_like(h)
    L = lib()
    L.d3d_oracle_gru_update.argtypes = [_f32p, _f32p, _f32p, ctypes.c_long, _f32p]
    L.d3d_oracle_gru_update(_p(u), _p(h), _p(convc), h.size, _p(out))
    return out


def groupnorm1(x, gamma, beta, eps=1e-5):
    """nn.GroupNorm(1, C) on x [C,...] (module.py:62-67)."""
    x = _c(x).copy()
    L = lib()
    L.d3d_oracle_groupnorm1.argtypes = [_f32p, ctypes.c_int, ctypes.c_long, _f32p, _f32p, ctypes.c_float]
    L.d3d_oracle_groupnorm1(_p(x), x.shape[0], x[0].size, _p(_c(gamma)), _p(_c(beta)), eps)
    return x


# ---- compositions restating the reference modules --------------------------------------

def conv_gru_cell(x, h, p, prefix):
    """module.py:24-51 ConvGRUCell.forward. p: dict of numpy weights, keys prefix+'conv_gates.0.weight' ..."""
    inp = np.concatenate([x, h], 0)
    gates = conv2d_k3(inp, p[prefix + "conv_gates.0.weight"], p[prefix + "conv_gates.0.bias"])
    H = h.shape[0]
    r = sigmoid(gates[:H])
    u = sigmoid(gates[H:])
    inp2 = np.concatenate([x, r * h], 0)
    c = conv2d_k3(inp2, p[prefix + "convc.0.weight"], p[prefix + "convc.0.bias"])
    return gru_update(u, h, c)


def slice_cost_reg_red(cost, s1, s2, p, prefix, up):
    """adamvs.py:418-427 SliceCostRegNetRED.forward (one depth slice)."""
    relu = lambda a: np.maximum(a, 0.0).astype(np.float32)
    c1 = relu(conv2d_k3(cost, p[prefix + "conv1.conv.weight"]))
    s1 = conv_gru_cell(c1, s1, p, prefix + "conv_gru1.")
    c2 = relu(conv2d_k3(s1, p[prefix + "conv2.conv.weight"], stride=2))
    s2 = conv_gru_cell(c2, s2, p, prefix + "conv_gru2.")
    up1 = convtranspose2d_k3s2(s2, p[prefix + "upconv1.weight"], p[prefix + "upconv1.bias"])
    up11 = relu(up1 + s1)
    if up:
        reg = convtranspose2d_k3s2(up11, p[prefix + "upconv2d.weight"], p[prefix + "upconv2d.bias"])
    else:
        reg = conv2d_k3(up11, p[prefix + "upconv2d.weight"], p[prefix + "upconv2d.bias"])
    return reg, s1, s2


def _cbr3d(x, p, prefix, stride=1, eps=1e-5):
    y = conv3d_k3(x, p[prefix + "conv.weight"], stride=stride)
    return bn_relu_add(y, p[prefix + "bn.weight"], p[prefix + "bn.bias"], p[prefix + "bn.running_mean"],
                       p[prefix + "bn.running_var"], eps, True)


def _ctbr3d(x, p, prefix, skip, eps=1e-5):
    y = convtranspose3d_k3s2(x, p[prefix + "0.weight"])
    return bn_relu_add(y, p[prefix + "1.weight"], p[prefix + "1.bias"], p[prefix + "1.running_mean"],
                       p[prefix + "1.running_var"], eps, True, skip)


def cost_reg_net_3d(x, p, prefix=""):
    """cas_mvsnet.py:112-121 CostRegNet.forward, eval-mode BN. x [C,D,H,W] -> [1,D,H,W]."""
    c0 = _cbr3d(x, p, prefix + "conv0.")
    c2 = _cbr3d(_cbr3d(c0, p, prefix + "conv1.", 2), p, prefix + "conv2.")
    c4 = _cbr3d(_cbr3d(c2, p, prefix + "conv3.", 2), p, prefix + "conv4.")
    x6 = _cbr3d(_cbr3d(c4, p, prefix + "conv5.", 2), p, prefix + "conv6.")
    x = _ctbr3d(x6, p, prefix + "conv7.", c4)
    x = _ctbr3d(x, p, prefix + "conv9.", c2)
    x = _ctbr3d(x, p, prefix + "conv11.", c0)
    return conv3d_k3(x, p[prefix + "prob.weight"], p[prefix + "prob.bias"])


def conv_gru_cell2(x, h, p, prefix):
    """module.py:53-99 ConvGRUCell2.forward."""
    H = h.shape[0]
    f = conv2d_k3(np.concatenate([x, h], 0), p[prefix + "gate_conv.weight"], p[prefix + "gate_conv.bias"])
    r = sigmoid(groupnorm1(f[:H], p[prefix + "reset_gate_norm.weight"], p[prefix + "reset_gate_norm.bias"]))
    u = sigmoid(groupnorm1(f[H:], p[prefix + "update_gate_norm.weight"], p[prefix + "update_gate_norm.bias"]))
    o = conv2d_k3(np.concatenate([x, r * h], 0), p[prefix + "output_conv.weight"], p[prefix + "output_conv.bias"])
    on = groupnorm1(o, p[prefix + "output_norm.weight"], p[prefix + "output_norm.bias"])
    return gru_update(u, h, on)


def slice_red_regularization(cost, states, p, prefix):
    """msrednet.py:353-370 slice_RED_Regularization.forward (one depth slice); states = [s1, s2, s3, s4]."""
    relu = lambda a: np.maximum(a, 0.0).astype(np.float32)
    s1, s2, s3, s4 = states
    neg = (-cost).astype(np.float32)
    c1 = relu(conv2d_k3(neg, p[prefix + "conv1.conv.weight"], stride=2))
    c2 = relu(conv2d_k3(c1, p[prefix + "conv2.conv.weight"], stride=2))
    c3 = relu(conv2d_k3(c2, p[prefix + "conv3.conv.weight"], stride=2))
    s4 = conv_gru_cell2(c3, s4, p, prefix + "conv_gru4.")
    up3 = relu(convtranspose2d_k3s2(s4, p[prefix + "upconv3.conv.weight"]))
    s3 = conv_gru_cell2(c2, s3, p, prefix + "conv_gru3.")
    up2 = relu(convtranspose2d_k3s2(up3 + s3, p[prefix + "upconv2.conv.weight"]))
    s2 = conv_gru_cell2(c1, s2, p, prefix + "conv_gru2.")
    up1 = relu(convtranspose2d_k3s2(up2 + s2, p[prefix + "upconv1.conv.weight"]))
    s1 = conv_gru_cell2(neg, s1, p, prefix + "conv_gru1.")
    # nn.ConvTranspose2d(8, 1, 3, stride=1, padding=1) == a correlation with the flipped, transposed kernel
    wt = p[prefix + "upconv2d.weight"]
    wc = np.ascontiguousarray(np.flip(wt, (2, 3)).transpose(1, 0, 2, 3))
    reg = conv2d_k3(up1 + s1, wc, p[prefix + "upconv2d.bias"])
    return reg, [s1, s2, s3, s4]
