"""Host-side mirror of the reference operator library (mvs/mvs_cas/models/module.py).

Same public names, argument order and meaning, and the same state_dict keys, so the
reference's Infer_* drivers and checkpoints carry over -- but the arithmetic of the hot
path runs in the gfx950 C-ABI kernels (deep3d_aerial_amd.ops).  torch.nn modules are used
as PARAMETER CONTAINERS (they define the checkpoint layout); their forward()s call the
HIP kernels.  The image feature pyramids are upstream of the plane-sweep path and
stay PyTorch-ROCm / MIOpen modules, except that their 3x3 layers reuse the matrix-core convolution.

Everything here requires GPU tensors: there is no CPU fallback.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import config as _cfg
from . import ops


# ----------------------------------------------------------------------------------------
# warp / depth hypotheses (module.py:516-557, 616-650)
# ----------------------------------------------------------------------------------------

def _feature_precision():
    """Feature pyramids run in exact fp32 whatever the regularisers' precision (D3D_FEATURE_PRECISION=follow: experiment that
    lets them follow the global mode)."""
    import contextlib

    return contextlib.nullcontext() if _cfg.get("D3D_FEATURE_PRECISION") == "follow" else ops.fp32_convs()


def compose_batch(proj_matrices):
    """[B,V,4,4] -> list of B tensors [V-1,12] (module.py:528-530 for every source view)."""
    return [ops.compose_projections(proj_matrices[b].contiguous()) for b in range(proj_matrices.shape[0])]


def homo_warping_float(src_fea, src_proj, ref_proj, depth_values):
    """module.py:516-557.  src_fea [B,C,H,W]; src_proj, ref_proj [B,4,4];
    depth_values [B,D] or [B,D,H,W]  ->  [B,C,D,H,W]."""
    B = src_fea.shape[0]
    outs = []
    for b in range(B):
        p34 = ops.compose_projections(torch.stack([ref_proj[b], src_proj[b]]).contiguous())
        outs.append(ops.homo_warp(src_fea[b].contiguous(), p34[0], depth_values[b].contiguous()))
    return torch.stack(outs)


def get_cur_depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, shape, max_depth=192.0, min_depth=0.0):
    """module.py:616-630. cur_depth [B,H,W] -> [B,D,H,W]. (max_depth/min_depth unused, as in the reference.)"""
    assert tuple(cur_depth.shape) == tuple(shape), "cur_depth:{}, input shape:{}".format(cur_depth.shape, shape)
    return torch.stack([ops.depth_range_samples(cur_depth[b].contiguous(), ndepth, float(depth_inteval_pixel))
                        for b in range(cur_depth.shape[0])])


def get_depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, device, dtype, shape, max_depth=192.0,
                            min_depth=0.0):
    """module.py:633-650. cur_depth [B,2]|(B,D) -> [B,D,H,W] tiled linspace; cur_depth [B,H,W] -> per-pixel."""
    if cur_depth.dim() == 2:
        planes = torch.stack([ops.depth_range_samples(cur_depth[b].contiguous(), ndepth, 0.0)
                              for b in range(cur_depth.shape[0])])
        return planes[:, :, None, None].expand(-1, -1, shape[1], shape[2]).contiguous()
    return get_cur_depth_range_samples(cur_depth, ndepth, depth_inteval_pixel, shape, max_depth, min_depth)


def plane_depths(cur_depth, ndepth):
    """Stage-1 hypotheses kept as [B,D] (one value per plane) instead of the tiled volume."""
    return torch.stack([ops.depth_range_samples(cur_depth[b].contiguous(), ndepth, 0.0)
                        for b in range(cur_depth.shape[0])])


# ----------------------------------------------------------------------------------------
# folded eval-mode BatchNorm
# ----------------------------------------------------------------------------------------
def folded_bn(bn):
    """(scale, shift) with y = x*scale + shift == eval-mode BatchNorm; cached per parameter version."""
    key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.device)
    cache = getattr(bn, "_d3d_fold", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
            shift = (bn.bias - bn.running_mean * scale).float().contiguous()
        ops.publish_prepared(scale)   # (read from every stream the forwards use)
        cache = (key, scale, shift)
        bn._d3d_fold = cache
    return cache[1], cache[2]


def _no_train(m):
    if m.training:
        raise RuntimeError("%s: the MI355X engine implements inference only (call .eval())" % type(m).__name__)


# ----------------------------------------------------------------------------------------
# 3D blocks (module.py:297-304)
# ----------------------------------------------------------------------------------------
class ConvBnReLU3D(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        assert kernel_size == 3 and pad == 1 and stride in (1, 2)
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm3d(out_channels)
        self.stride = stride

    def forward(self, x, skip=None):  # x [C,D,H,W] (unbatched)
        _no_train(self)
        s, t = folded_bn(self.bn)
        return ops.conv3d_k3(x, self.conv.weight, s, t, skip, relu=True, stride=self.stride)

    def forward_cl(self, x, skip=None):  # bf16 mode: x planar fp32 or channel-last bf16 -> channel-last bf16
        _no_train(self)
        s, t = folded_bn(self.bn)
        return ops.conv3d_k3_cl(x, self.conv.weight, s, t, skip, relu=True, stride=self.stride)


# ----------------------------------------------------------------------------------------
# 2D blocks used by the regularisers (module.py:248-274, 5-51)
# ----------------------------------------------------------------------------------------
class ConvBnReLU(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        assert kernel_size == 3 and pad == 1 and stride in (1, 2)
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)
        self.stride = stride

    def forward(self, x):  # x [C,H,W]
        _no_train(self)
        s, t = folded_bn(self.bn)
        return ops.conv2d_k3(x, self.conv.weight, s, t, None, act=1, stride=self.stride)


class ConvReLU(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        assert kernel_size == 3 and pad == 1 and stride in (1, 2)
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.stride = stride

    def forward(self, x):
        return ops.conv2d_k3(x, self.conv.weight, None, None, None, act=1, stride=self.stride)


class ConvGRUCell(nn.Module):
    """module.py:5-51.  forward(x [Cx,H,W], h [Ch,H,W]) -> (h', h')."""

    def __init__(self, input_channels, hidden_channels, kernel_size=3):
        super().__init__()
        assert kernel_size == 3
        self.input_channels = input_channels
        self.hidden_channels = hidden_channels
        cin = input_channels + hidden_channels
        self.conv_gates = nn.Sequential(nn.Conv2d(cin, 2 * hidden_channels, 3, stride=1, padding=1, bias=True))
        self.convc = nn.Sequential(nn.Conv2d(cin, hidden_channels, 3, stride=1, padding=1, bias=True))

    def forward(self, x, h):
        if h is None:
            h = torch.zeros((self.hidden_channels,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
        g = self.conv_gates[0]
        fused = ops.gru_cell_fused(x, h, g.weight, g.bias, self.convc[0].weight, self.convc[0].bias)
        if fused is not None:
            return fused, fused
        gates = ops.conv2d_k3(x, g.weight, None, g.bias, None, act=0, stride=1, x2=h)
        rh, u = ops.gru_gates(gates, h)
        c = self.convc[0]
        cand = ops.conv2d_k3(x, c.weight, None, c.bias, None, act=0, stride=1, x2=rh)
        out = ops.gru_update(u, h, cand)
        return out, out


class ConvTransReLU(nn.Module):
    """module.py:287-294 (k=3, stride 2, pad 1, output_pad 1, no bias): relu(convT(x)) [+ skip, added after]."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=2, pad=1, output_pad=1):
        super().__init__()
        assert kernel_size == 3 and stride == 2 and pad == 1 and output_pad == 1
        self.conv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=3, stride=2, padding=1, output_padding=1,
                                       bias=False)

    def forward(self, x, skip=None):
        return ops.convtranspose2d_k3s2(x, self.conv.weight, None, None, skip, skip_after_act=True, act=1)


class ConvGRUCell2(nn.Module):
    """module.py:53-99: conv-GRU whose gate and candidate convolutions are followed by GroupNorm(1, C).
    forward(x [Cx,H,W], h [Ch,H,W], negate_x=False) -> (h', h').  negate_x evaluates the cell on -x by negating
    the x-columns of the two weight tensors (msrednet.py:354,367 feed -cost)."""

    def __init__(self, input_channel, output_channel, kernel_size=3):
        super().__init__()
        assert kernel_size == 3
        self.input_channel, self.output_channel = input_channel, output_channel
        cin = input_channel + output_channel
        self.gate_conv = nn.Conv2d(cin, output_channel * 2, 3, padding=1)
        self.reset_gate_norm = nn.GroupNorm(1, output_channel, 1e-5, True)
        self.update_gate_norm = nn.GroupNorm(1, output_channel, 1e-5, True)
        self.output_conv = nn.Conv2d(cin, output_channel, 3, padding=1)
        self.output_norm = nn.GroupNorm(1, output_channel, 1e-5, True)

    def _w(self, conv, negate_x):
        if not negate_x:
            return conv.weight
        ci = self.input_channel
        return ops.derived_weight(conv.weight, "negx", lambda w: torch.cat([-w[:, :ci], w[:, ci:]], 1))

    def forward(self, x, h=None, negate_x=False):
        if h is None:
            h = torch.zeros((self.output_channel,) + tuple(x.shape[1:]), dtype=torch.float32, device=x.device)
        rn, un, on = self.reset_gate_norm, self.update_gate_norm, self.output_norm
        if x.is_cuda:   # fast mode: the cell's four launches issued by ONE library call (ops.gru2_cell_gn); None: layer by layer below
            out = ops.gru2_cell_gn(x, h, self._w(self.gate_conv, negate_x), self.gate_conv.bias, self._w(self.output_conv, negate_x),
                                   self.output_conv.bias, rn, un, on)
            if out is not None:
                return out, out
        # (the GroupNorm statistics ride on the convolutions' epilogues where the kernel has that form: ops.GnStats)
        gs = ops.GnStats(2)
        f = ops.conv2d_k3(x, self._w(self.gate_conv, negate_x), None, self.gate_conv.bias, None, act=0, stride=1, x2=h, gn=gs)
        rn, un = self.reset_gate_norm, self.update_gate_norm
        st = gs.stats(f)
        # the reset half alone, the update gate evaluated inside the state update (104 channel planes per cell instead of 128, the
        # same bits): ops.gru_reset_gn / gru_update_gates_gn; the two-output form where the kernel does not take the shape
        rh = None
        if f.is_cuda and rn.eps == un.eps == self.output_norm.eps and not _cfg.off("gru_gates_split"):
            rh = ops.gru_reset_gn(f, h, rn.weight, rn.bias, rn.eps, st[0])
        if rh is None:
            rh, u = ops.gru_gates_gn(f, h, rn.weight, rn.bias, un.weight, un.bias, rn.eps, stats=st)
        else:
            u = None
        go = ops.GnStats(1)
        o = ops.conv2d_k3(x, self._w(self.output_conv, negate_x), None, self.output_conv.bias, None, act=0, stride=1,
                          x2=rh, gn=go)
        on = self.output_norm
        if u is None:
            out = ops.gru_update_gates_gn(o, f, h, on.weight, on.bias, un.weight, un.bias, on.eps, go.stats(o), st[1])
        else:
            out = ops.gru_update_gn(o, u, h, on.weight, on.bias, on.eps, stats=go.stats(o))
        return out, out


# ----------------------------------------------------------------------------------------
# Image feature pyramids -- PyTorch-ROCm/MIOpen, checkpoint-compatible
# with module.py:157-245,495-513,653-755.
# ----------------------------------------------------------------------------------------
def feature_conv(conv, x, bn=None, relu=False, skip=None, x2=None):
    """nn.Conv2d `conv` (+ eval-mode BatchNorm `bn`, + ReLU, + `skip` added last) on a batch [B,C,H,W]; `x2`
    [B,C2,H,W] is a second input concatenated after x along the channels without materialising the concat.  Odd square kernels with
    padding K // 2 and stride 1 | 2 (the 3x3, 5x5 stride-2 and 1x1 layers of the feature pyramids) run on the
    matrix-core convolution in exact fp32; anything else, or D3D_FEATURE_CONV=miopen, goes to MIOpen."""
    K = conv.kernel_size[0]
    if (conv.kernel_size == (K, K) and K % 2 == 1 and conv.padding == (K // 2, K // 2) and conv.stride in ((1, 1), (2, 2))
            and conv.dilation == (1, 1) and conv.groups == 1 and x.is_cuda and x.dtype == torch.float32
            and conv.out_channels <= 64 and (bn is None or not bn.training)
            and _cfg.get("D3D_FEATURE_CONV") != "miopen"):
        if bn is not None:
            s, t = folded_bn(bn)
        else:
            s, t = None, conv.bias
        with _feature_precision():  # the bf16 mode (BASELINE config 3) is for the cost regularisers only
            outs = []
            for b in range(x.shape[0]):
                xb = x[b].contiguous()
                sk = None if skip is None else skip[b].contiguous()
                if K == 3:
                    y = ops.conv2d_k3(xb, conv.weight, s, t, sk, act=1 if relu else 0, stride=conv.stride[0],
                                      x2=None if x2 is None else x2[b].contiguous())
                elif x2 is None:
                    y = ops.conv2d_same(xb, conv.weight, s, t, sk, act=1 if relu else 0, stride=conv.stride[0])
                else:
                    y = None
                if y is None:
                    break
                outs.append(y)
            else:
                return outs[0].unsqueeze(0) if len(outs) == 1 else torch.stack(outs)
    if x2 is not None:
        x = torch.cat((x, x2), dim=1)
    y = conv(x)
    if bn is not None:
        y = bn(y)
    y = F.relu(y, inplace=True) if relu else y
    return y if skip is None else y + skip


def lateral_upsample_add(conv, x, coarse):
    """`F.interpolate(coarse, scale_factor=2, mode="nearest") + conv(x)` (module.py:742,746 of the reference) with the
    upsampling folded into the 1x1 convolution's epilogue where the fused kernel takes the shape."""
    if (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.groups == 1 and x.is_cuda and x.dtype == torch.float32
            and _cfg.get("D3D_FEATURE_CONV") != "miopen"):
        outs = []
        for b in range(x.shape[0]):
            y = ops.conv1x1_upskip(x[b].contiguous(), conv.weight, conv.bias, coarse[b].contiguous())
            if y is None:
                break
            outs.append(y)
        else:
            return outs[0].unsqueeze(0) if len(outs) == 1 else torch.stack(outs)
    return feature_conv(conv, x, skip=F.interpolate(coarse, scale_factor=2, mode="nearest"))


def _fpn_weights(lateral, head):
    """Composite operands of fpn_output (host-side weight preparation, cached per parameter version)."""
    w1 = lateral.weight
    tag = "_%d_%d" % (w1.data_ptr(), w1._version) + ("" if lateral.bias is None else "_%d_%d" % (lateral.bias.data_ptr(), lateral.bias._version))

    def comp(w3):   # (W3 . W1): a 3x3 convolution straight from the lateral's input channels
        return torch.einsum("omyx,mc->ocyx", w3.double(), w1.detach().double().reshape(w1.shape[0], -1)).float()

    def bias_taps(w3):   # what the lateral bias contributes through each tap: [Co,3,3], and its sum over the taps [Co]
        bt = torch.einsum("omyx,m->oyx", w3.double(), lateral.bias.detach().double()).float()
        return bt, bt.sum((1, 2))

    wt = ops.derived_weight(head.weight, "fpn_up", ops.upsampled_conv_weight)
    wb = ops.derived_weight(head.weight, "fpn_lat" + tag, comp)
    bt, bsum = (None, None) if lateral.bias is None else ops.derived_weight(head.weight, "fpn_bias" + tag, bias_taps)
    return wt, wb, bt, bsum


def fpn_output(lateral, x, coarse, head, wide=None):
    """`head(F.interpolate(coarse, x2, nearest) + lateral(x))` (module.py:745-747 of the reference) without the wide tensor at
    the output resolution.  The layer is linear in its input: head(up(coarse)) is a ConvTranspose2d(k 4, s 2, p 1) of
    `coarse` with summed weights, head(lateral(x)) one 3x3 convolution of x with the composite weights W3 . W1, and the
    lateral bias reaches the output through the taps that lie inside the image (a constant, corrected on the one-pixel
    border).  Falls back to the two-kernel path for shapes the tile kernels do not take."""
    Co, Cm = head.out_channels, head.in_channels
    if (lateral.kernel_size == (1, 1) and lateral.groups == 1 and head.kernel_size == (3, 3) and head.padding == (1, 1)
            and head.stride == (1, 1) and head.dilation == (1, 1) and head.groups == 1 and x.is_cuda and x.dtype == torch.float32
            and x.shape[1] in (8, 16) and Cm == 32 and Co <= 16 and coarse.shape[1] == Cm and x.shape[3] % 8 == 0
            and x.shape[2] == 2 * coarse.shape[2] and x.shape[3] == 2 * coarse.shape[3] and _feature_precision_is_fp32()
            and _cfg.get("D3D_FEATURE_CONV") != "miopen" and not _cfg.off("fpn_split")):
        wt, wb, bt, bsum = _fpn_weights(lateral, head)   # bt and its sum come from ONE cache entry keyed on both layers' versions
        bias = head.bias if bt is None else bsum if head.bias is None else bsum + head.bias
        outs = []
        with _feature_precision():
            for b in range(x.shape[0]):
                lat = ops.conv2d_zs(x[b].contiguous(), wb, None, bias)
                if lat is None:
                    break
                if bt is not None:   # taps outside the image carry no bias: the one-pixel border
                    ops.conv3x3_bias_border_(lat, bt)
                y = ops.convtranspose2d_k4_zs(coarse[b].contiguous(), wt, None, None, lat)
                if y is None:
                    break
                outs.append(y)
            else:
                return outs[0].unsqueeze(0) if len(outs) == 1 else torch.stack(outs)
    return feature_conv(head, lateral_upsample_add(lateral, x, coarse) if wide is None else wide)   # wide: the sum, where the caller has it


def _feature_precision_is_fp32():
    return _cfg.get("D3D_FEATURE_PRECISION") != "follow" or ops.conv_precision() != "h16"


class Conv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, relu=True, bn=True, bn_momentum=0.1,
                 **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn), **kwargs)
        self.bn = nn.BatchNorm2d(out_channels, momentum=bn_momentum) if bn else None
        self.relu = relu

    def forward(self, x, x2=None):
        return feature_conv(self.conv, x, self.bn, self.relu, x2=x2)


class Deconv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, relu=True, bn=True, bn_momentum=0.1,
                 **kwargs):
        super().__init__()
        assert stride in (1, 2)
        self.stride = stride
        self.conv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride=stride, bias=(not bn),
                                       **kwargs)
        self.bn = nn.BatchNorm2d(out_channels, momentum=bn_momentum) if bn else None
        self.relu = relu

    def forward(self, x):
        c = self.conv
        if (self.stride == 2 and c.kernel_size == (3, 3) and c.padding == (1, 1) and c.output_padding == (1, 1)
                and c.out_channels <= 64 and x.is_cuda and (self.bn is None or not self.bn.training)
                and _cfg.get("D3D_FEATURE_CONV") != "miopen"):
            s, t = folded_bn(self.bn) if self.bn is not None else (None, c.bias)
            with _feature_precision():
                outs = [ops.convtranspose2d_k3s2(x[b].contiguous(), c.weight, s, t, None, act=1 if self.relu else 0)
                        for b in range(x.shape[0])]
            return outs[0].unsqueeze(0) if len(outs) == 1 else torch.stack(outs)
        y = self.conv(x)
        if self.stride == 2:
            y = y[:, :, :2 * x.shape[2], :2 * x.shape[3]].contiguous()
        if self.bn is not None:
            y = self.bn(y)
        return F.relu(y, inplace=True) if self.relu else y


class DeConv2dFuse(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, relu=True, bn=True, bn_momentum=0.1):
        super().__init__()
        self.deconv = Deconv2d(in_channels, out_channels, kernel_size, stride=2, padding=1, output_padding=1,
                               bn=True, relu=relu, bn_momentum=bn_momentum)
        self.conv = Conv2d(2 * out_channels, out_channels, kernel_size, stride=1, padding=1, bn=bn, relu=relu,
                           bn_momentum=bn_momentum)

    def forward(self, x_pre, x):
        return self.conv(self.deconv(x), x2=x_pre)  # conv over cat((deconv(x), x_pre), 1) without the concat


def trunk_conv0(conv0, x):
    """conv0 of a feature trunk (two ConvBnReLU at full resolution, module.py:663-666) on a batch [B,3,H,W]: one launch per image
    where ops.conv2d_k3_pair3 takes the shape (the 8-channel map between the layers is never written), else layer by layer."""
    a, b = conv0[0], conv0[1]
    if (x.is_cuda and x.dtype == torch.float32 and x.shape[1] == 3 and a.bn is not None and b.bn is not None
            and not a.bn.training and not b.bn.training and _cfg.get("D3D_FEATURE_CONV") != "miopen"):
        s0, t0 = folded_bn(a.bn)
        s1, t1 = folded_bn(b.bn)
        with _feature_precision():
            outs = []
            for i in range(x.shape[0]):
                y = ops.conv2d_k3_pair3(x[i].contiguous(), a.conv.weight, s0, t0, 1 if a.relu else 0,
                                        b.conv.weight, s1, t1, 1 if b.relu else 0)
                if y is None:
                    break
                outs.append(y)
            else:
                return outs[0].unsqueeze(0) if len(outs) == 1 else torch.stack(outs)
    return conv0(x)


def _trunk(base):
    """conv0/conv1/conv2 of both feature nets (module.py:663-679, adamvs.py:59-75)."""
    conv0 = nn.Sequential(Conv2d(3, base, 3, 1, padding=1), Conv2d(base, base, 3, 1, padding=1))
    conv1 = nn.Sequential(Conv2d(base, base * 2, 5, stride=2, padding=2), Conv2d(base * 2, base * 2, 3, 1, padding=1),
                          Conv2d(base * 2, base * 2, 3, 1, padding=1))
    conv2 = nn.Sequential(Conv2d(base * 2, base * 4, 5, stride=2, padding=2),
                          Conv2d(base * 4, base * 4, 3, 1, padding=1), Conv2d(base * 4, base * 4, 3, 1, padding=1))
    return conv0, conv1, conv2


class FeatureNet_mvsnet(nn.Module):
    """module.py:653-755 (the 3-stage 'fpn' and 'unet' variants used at inference)."""

    def __init__(self, base_channels, num_stage=3, stride=4, arch_mode="unet"):
        super().__init__()
        assert arch_mode in ("unet", "fpn") and num_stage == 3
        self.arch_mode, self.stride, self.base_channels, self.num_stage = arch_mode, stride, base_channels, num_stage
        b = base_channels
        self.conv0, self.conv1, self.conv2 = _trunk(b)
        self.out1 = nn.Conv2d(b * 4, b * 4, 1, bias=False)
        if arch_mode == "unet":
            self.deconv1 = DeConv2dFuse(b * 4, b * 2, 3)
            self.deconv2 = DeConv2dFuse(b * 2, b, 3)
            self.out2 = nn.Conv2d(b * 2, b * 2, 1, bias=False)
            self.out3 = nn.Conv2d(b, b, 1, bias=False)
        else:
            self.inner1 = nn.Conv2d(b * 2, b * 4, 1, bias=True)
            self.inner2 = nn.Conv2d(b, b * 4, 1, bias=True)
            self.out2 = nn.Conv2d(b * 4, b * 2, 3, padding=1, bias=False)
            self.out3 = nn.Conv2d(b * 4, b, 3, padding=1, bias=False)
        self.out_channels = [4 * b, 2 * b, b]

    def forward(self, x):
        c0 = trunk_conv0(self.conv0, x)
        c1 = self.conv1(c0)
        c2 = self.conv2(c1)
        out = {"stage1": feature_conv(self.out1, c2)}
        if self.arch_mode == "unet":
            f = self.deconv1(c1, c2)
            out["stage2"] = feature_conv(self.out2, f)
            f = self.deconv2(c0, f)
            out["stage3"] = feature_conv(self.out3, f)
        else:
            f = lateral_upsample_add(self.inner1, c1, c2)   # (the coarse input of the last level)
            out["stage2"] = fpn_output(self.inner1, c1, c2, self.out2, wide=f)
            out["stage3"] = fpn_output(self.inner2, c0, f, self.out3)
        return out
