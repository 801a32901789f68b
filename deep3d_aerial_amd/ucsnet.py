"""UCS-Net inference on the MI355X plane-sweep engine: variance cost volume, 3D-UNet regulariser, soft-argmin with the
spread of the distribution, uncertainty-aware hypotheses for the next stage.

Mirror of the reference's mvs/mvs_cas/models/ucsnet.py (class name, constructor arguments, forward() contract,
state_dict keys).  `Infer_UCSNet` cannot be built by the reference's own harness -- predict.py:80 passes `num_depth=`,
which the constructor does not take, and forward reads `self.num_depth`, which nothing sets.  Here the
constructor accepts `num_depth` (stored; the value only feeds an interval the reference never uses), so the harness call
works; the goldens come from the reference class built with its own arguments and that one attribute assigned.

    ops.compose_projections        module.py:528-530
    ops.variance_volume            ucsnet.py:119-134   (fused warp + variance)
    CostRegNet                     ucsnet.py:56-96     (the CasMVSNet regulariser, same keys)
    ops.softargmin_conf4_var       ucsnet.py:137-151
    ops.uncertainty_aware_samples  ucsnet.py:30-53
    ops.resize_bilinear            ucsnet.py:283-284   (F.interpolate bilinear of depth and variance)
"""
import torch
import torch.nn as nn

from . import ops
from .cas_mvsnet import CostRegNet
from .dataset import extract_features
from .module import FeatureNet_mvsnet


# Stages 2 and 3 hand the sweep and the regression the two maps (low, step) their hypotheses are generated from instead of the
# [D,h,w] volume (ops.AffineDepth, as cas_mvsnet.AFFINE_DEPTH): no D-plane volume is written or read, the window kernel bounds its
# windows from two maps.  False restores the volumes.
AFFINE_DEPTH = True


def compute_depth(feats, proj_mats, depth_samps, cost_reg, lamb):
    """ucsnet.py:99-151 for one batch item: feats list of [C,h,w], proj_mats [V,4,4], depth_samps [D] or [D,h,w]."""
    assert len(feats) == proj_mats.shape[0], "Different number of images and projection matrices"
    p34 = ops.compose_projections(proj_mats.contiguous())
    cl = cost_reg.channel_last() and feats[0].shape[0] % 8 == 0
    var = ops.variance_volume_cl(feats, p34, depth_samps, layout="cl8") if cl else ops.variance_volume(feats, p34, depth_samps)
    cost = cost_reg.forward_one(var)
    return ops.softargmin_conf4_var(cost, depth_samps, lamb)


class Infer_UCSNet(nn.Module):
    """ucsnet.py:234-311.  forward(imgs [B,V,3,H,W], proj_matrices {stageN: [B,V,4,4]}, depth_values [B,2+])."""

    def __init__(self, lamb=1.5, ndepths=[64, 32, 8], grad_method="detach", arch_mode="unet", base_chs=[8, 8, 8],
                 num_depth=None):
        super().__init__()
        self.ndepths, self.grad_method, self.arch_mode, self.base_chs = list(ndepths), grad_method, arch_mode, list(base_chs)
        self.lamb, self.num_stage, self.num_depth = lamb, len(ndepths), num_depth
        self.stage_infos = {"stage1": 4.0, "stage2": 2.0, "stage3": 1.0}
        self.feature_extraction = FeatureNet_mvsnet(base_channels=8, stride=4, num_stage=self.num_stage, arch_mode=arch_mode)
        self.cost_regularization = nn.ModuleList([CostRegNet(in_channels=self.feature_extraction.out_channels[i],
                                                             base_channels=self.base_chs[i]) for i in range(self.num_stage)])
        self.feature_cache = None   # dataset.FeatureCache: pyramids of shared images across reference views (predict_views)

    def forward(self, imgs, proj_matrices, depth_values, image_keys=None):
        """image_keys (optional, with self.feature_cache set): one hashable key per view, as in the other drivers."""
        features = extract_features(self.feature_extraction, imgs, image_keys, self.feature_cache)
        B, _, H, W = features[0]["stage3"].shape   # the finest level has the image's size
        outputs = {}
        depth, exp_var = None, None
        for stage_idx in range(self.num_stage):
            key = "stage{}".format(stage_idx + 1)
            feats = [f[key] for f in features]
            scale = int(self.stage_infos[key])
            cur_h, cur_w = H // scale, W // scale
            deps, confs, variances = [], [], []
            for b in range(B):
                if depth is None:
                    samples = ops.uncertainty_aware_samples(depth_values[b].contiguous(), None, self.ndepths[stage_idx])
                else:
                    cur = ops.resize_bilinear(depth[b:b + 1].contiguous(), cur_h, cur_w)[0]
                    var = ops.resize_bilinear(exp_var[b:b + 1].contiguous(), cur_h, cur_w)[0]
                    samples = ops.uncertainty_aware_samples(cur, var, self.ndepths[stage_idx], affine=AFFINE_DEPTH)
                d, c, v = compute_depth([f[b].contiguous() for f in feats], proj_matrices[key][b], samples,
                                        self.cost_regularization[stage_idx], self.lamb)
                deps.append(d)
                confs.append(c)
                variances.append(v)
            depth, exp_var = torch.stack(deps), torch.stack(variances)
            stage_out = {"depth": depth, "photometric_confidence": torch.stack(confs), "variance": exp_var}
            outputs[key] = stage_out
            outputs.update(stage_out)
        return outputs
