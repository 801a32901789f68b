"""CasMVSNet inference on the MI355X plane-sweep engine: variance cost volume, 3D-UNet
regulariser, softmax / soft-argmin / 4-plane confidence.

Mirror of the reference's mvs/mvs_cas/models/cas_mvsnet.py (class names, constructor
arguments, forward() contract, state_dict keys).  Hot path per stage:

    ops.compose_projections   module.py:528-530
    ops.variance_volume       cas_mvsnet.py:45-60   (fused warp + variance, one kernel)
    CostRegNet                cas_mvsnet.py:81-121  (ops.conv3d_k3 / convtranspose3d_k3s2)
    ops.softargmin_conf4      cas_mvsnet.py:69-76
"""
import torch
import torch.nn as nn

from .dataset import extract_features
from . import ops
from .module import ConvBnReLU3D, FeatureNet_mvsnet, folded_bn, plane_depths, _no_train


class _Up3D(nn.Sequential):
    """ConvTranspose3d + BatchNorm3d + ReLU as indices 0,1,2 (cas_mvsnet.py:94-108 key layout)."""

    def __init__(self, cin, cout):
        super().__init__(nn.ConvTranspose3d(cin, cout, kernel_size=3, padding=1, output_padding=1, stride=2,
                                            bias=False),
                         nn.BatchNorm3d(cout), nn.ReLU(inplace=True))

    def forward(self, x, skip):
        _no_train(self)
        s, t = folded_bn(self[1])
        return ops.convtranspose3d_k3s2(x, self[0].weight, s, t, skip, relu=True)

    def forward_cl(self, x, skip):  # channel-last bf16 volumes (ops.conv3d_k3_cl)
        _no_train(self)
        s, t = folded_bn(self[1])
        return ops.convtranspose3d_k3s2_cl(x, self[0].weight, s, t, skip, relu=True)


class CostRegNet(nn.Module):
    """cas_mvsnet.py:81-121.  forward(x [B,C,D,H,W]) -> [B,1,D,H,W]."""

    def __init__(self, in_channels, base_channels=8):
        super().__init__()
        self.conv0 = ConvBnReLU3D(in_channels, 8)
        self.conv1 = ConvBnReLU3D(8, 16, stride=2)
        self.conv2 = ConvBnReLU3D(16, 16)
        self.conv3 = ConvBnReLU3D(16, 32, stride=2)
        self.conv4 = ConvBnReLU3D(32, 32)
        self.conv5 = ConvBnReLU3D(32, 64, stride=2)
        self.conv6 = ConvBnReLU3D(64, 64)
        self.conv7 = _Up3D(64, 32)
        self.conv9 = _Up3D(32, 16)
        self.conv11 = _Up3D(16, 8)
        self.prob = nn.Conv3d(8, 1, 3, stride=1, padding=1)

    @staticmethod
    def channel_last():
        """bf16 mode with channel-last bf16 activations between the layers (ops.conv3d_k3_cl)."""
        return ops.conv_precision() == "h16" and ops.channel_last_enabled()

    def forward_one(self, x):  # [C,D,H,W] fp32 (or channel-last bf16 [D,H,W,C] in bf16 mode) -> [D,H,W]
        cl_in = x.dtype == ops.h16_dtype()   # [D,H,W,C], or the sweep kernels' CL8 form [D,C/8,H,W,8]
        D, H, W = ((x.shape[0], x.shape[2], x.shape[3]) if x.dim() == 5 else x.shape[:3]) if cl_in else x.shape[1:]
        if D % 8 or H % 8 or W % 8:
            raise ValueError("CostRegNet needs D,H,W divisible by 8 (got %s)" % ((D, H, W),))
        if cl_in or (self.channel_last() and x.shape[0] % 8 == 0):
            # bf16 mode (BASELINE config 3): every layer rounds its operands to bf16 anyway, so the activations travel
            # between the layers as channel-last bf16 volumes (half the traffic, 16-byte staging loads)
            c0 = self.conv0.forward_cl(x)
            c2 = self.conv2.forward_cl(self.conv1.forward_cl(c0))
            c4 = self.conv4.forward_cl(self.conv3.forward_cl(c2))
            y = self.conv6.forward_cl(self.conv5.forward_cl(c4))
            y = self.conv7.forward_cl(y, c4)
            y = self.conv9.forward_cl(y, c2)
            _no_train(self.conv11)
            s11, t11 = folded_bn(self.conv11[1])
            fused = ops.convtranspose3d_prob_cl(y, self.conv11[0].weight, s11, t11, c0, self.prob.weight, self.prob.bias)
            if fused is not None:   # conv11 + prob in one kernel: the full-resolution 8-channel volume stays in LDS
                return fused
            y = self.conv11.forward_cl(y, c0)
            return ops.conv3d_k3_cl(y, self.prob.weight, None, self.prob.bias, None, relu=False, stride=1, out_cl=False)[0]
        c0 = self.conv0(x)
        c2 = self.conv2(self.conv1(c0))
        c4 = self.conv4(self.conv3(c2))
        y = self.conv6(self.conv5(c4))
        y = self.conv7(y, c4)
        y = self.conv9(y, c2)
        y = self.conv11(y, c0)
        return ops.conv3d_k3(y, self.prob.weight, None, self.prob.bias, None, relu=False, stride=1)[0]

    def forward(self, x):
        return torch.stack([self.forward_one(x[b].contiguous()) for b in range(x.shape[0])]).unsqueeze(1)


CL_LAYOUT = "cl8"   # layout of the variance volume handed to conv0 in bf16 mode ("cl": [D,h,w,C], rounds 2-3)
AFFINE_DEPTH = True   # stages 2+ hand the sweep (lo, step) maps instead of a [D,h,w] hypothesis volume (ops.AffineDepth)


class DepthNet(nn.Module):
    """cas_mvsnet.py:31-78."""

    def forward(self, features, proj_matrices, depth_values, num_depth, cost_regularization, prob_volume_init=None):
        assert len(features) == proj_matrices.shape[1], "Different number of images and projection matrices"
        affine = isinstance(depth_values, (list, tuple))   # one ops.AffineDepth per batch item (see Infer_CascadeMVSNet.forward)
        nd = depth_values[0].D if affine else depth_values.shape[1]
        assert nd == num_depth, "depth_values.shape[1]:{}  num_depth:{}".format(nd, num_depth)
        if prob_volume_init is not None:
            raise NotImplementedError("prob_volume_init is never passed at inference (cas_mvsnet.py:228-230)")
        depths, confs = [], []
        for b in range(features[0].shape[0]):
            p34 = ops.compose_projections(proj_matrices[b].contiguous())
            dv = depth_values[b] if affine else depth_values[b].contiguous()
            fb = [f[b].contiguous() for f in features]
            # bf16 mode: the volume leaves the sweep kernel in the form conv0 stages (same rounding, half the bytes)
            cl = getattr(cost_regularization, "channel_last", None)
            # (CL8: planes of 8-channel groups -- whole-cell stores for the 16- and 32-channel stages)
            var = ops.variance_volume_cl(fb, p34, dv, layout=CL_LAYOUT) if (cl is not None and cl() and fb[0].shape[0] % 8 == 0) \
                else ops.variance_volume(fb, p34, dv)
            cost = cost_regularization.forward_one(var)
            d, c = ops.softargmin_conf4(cost, dv)
            depths.append(d)
            confs.append(c)
        return {"depth": torch.stack(depths), "photometric_confidence": torch.stack(confs)}


class Infer_CascadeMVSNet(nn.Module):
    """cas_mvsnet.py:140-241.  forward(imgs [B,V,3,H,W], proj_matrices {stageN: [B,V,4,4]}, depth_values [B,2])."""

    def __init__(self, refine=False, num_depth=384, ndepths=[48, 32, 8], depth_intervals_ratio=[4, 2, 1],
                 share_cr=False, grad_method="detach", arch_mode="fpn", cr_base_chs=[8, 8, 8]):
        super().__init__()
        if refine:
            raise NotImplementedError("RefineNet is broken in the reference (F.cat) and disabled there")
        assert len(ndepths) == len(depth_intervals_ratio)
        self.refine, self.share_cr, self.ndepths = refine, share_cr, list(ndepths)
        self.depth_intervals_ratio, self.grad_method, self.arch_mode = list(depth_intervals_ratio), grad_method, arch_mode
        self.cr_base_chs, self.num_stage, self.num_depth = list(cr_base_chs), len(ndepths), num_depth
        self.stage_infos = {"stage1": {"scale": 4.0}, "stage2": {"scale": 2.0}, "stage3": {"scale": 1.0}}
        self.feature = FeatureNet_mvsnet(base_channels=8, stride=4, num_stage=self.num_stage, arch_mode=arch_mode)
        if share_cr:
            self.cost_regularization = CostRegNet(in_channels=self.feature.out_channels[1], base_channels=8)
        else:
            self.cost_regularization = nn.ModuleList(
                [CostRegNet(in_channels=self.feature.out_channels[i], base_channels=self.cr_base_chs[i])
                 for i in range(self.num_stage)])
        self.DepthNet = DepthNet()

    feature_cache = None  # dataset.FeatureCache shared across reference views (set by the harness); see image_keys

    def forward(self, imgs, proj_matrices, depth_values, image_keys=None):
        """image_keys (optional, with self.feature_cache set): one hashable key per view; the feature pyramid of a
        key seen before is reused instead of recomputed, and imgs may then be a list whose cached entries are None."""
        # one device sync, like the reference's depth_values[0,0].cpu() (cas_mvsnet.py:184-185)
        dmin, dmax = ops.depth_range_host(depth_values)   # (host numbers: no device access when the caller noted them)
        depth_interval = (dmax - dmin) / self.num_depth

        features = extract_features(self.feature, imgs, image_keys, self.feature_cache)
        V = len(features)
        B, _, img_h, img_w = features[0]["stage3"].shape  # the finest level has the image's size
        outputs = {}
        depth = None
        for s in range(self.num_stage):
            key = "stage%d" % (s + 1)
            feats = [f[key] for f in features]
            scale = int(self.stage_infos[key]["scale"])
            h, w = img_h // scale, img_w // scale
            D = self.ndepths[s]
            if depth is None:
                dv = plane_depths(depth_values, D)  # [B,D]
            else:
                # cas_mvsnet.py:211-226: depth -> full res (bilinear), hypotheses at full res,
                # then the trilinear resample to the stage grid (identity along D).
                # The hypotheses are affine in the plane index (module.py:616-631: lo + k * step per pixel) and the resample
                # is bilinear plane by plane, so the two generating maps travel instead of the [D,h,w] volume: the sweep
                # and the soft-argmin read 2 maps, not D planes (AFFINE_DEPTH = False: the volume, as the reference).
                dvs = []
                for b in range(B):
                    cur = ops.resize_bilinear(depth[b:b + 1].contiguous(), img_h, img_w)[0]
                    if AFFINE_DEPTH:
                        aff = ops.depth_range_affine(cur, D, self.depth_intervals_ratio[s] * depth_interval)
                        dvs.append(aff if (h, w) == (img_h, img_w) else ops.AffineDepth(ops.resize_bilinear(aff.maps, h, w), D))
                    else:
                        full = ops.depth_range_samples(cur, D, self.depth_intervals_ratio[s] * depth_interval)
                        dvs.append(full if (h, w) == (img_h, img_w) else ops.resize_bilinear(full, h, w))
                dv = dvs if AFFINE_DEPTH else torch.stack(dvs)
            cr = self.cost_regularization if self.share_cr else self.cost_regularization[s]
            out = self.DepthNet(feats, proj_matrices[key], depth_values=dv, num_depth=D, cost_regularization=cr)
            depth = out["depth"]
            outputs[key] = out
            outputs.update(out)
        return outputs
