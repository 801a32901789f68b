"""Full-size check of the channel-last sweep output against the planar one (stage-2 shape of the cascade)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import predict, ops, synthetic as S
s = predict.SyntheticBlock(1, 5, 2752, 1856, 384, seed=9)[0]
for stage, C, D, sc in (("stage2", 16, 32, 2), ("stage1", 32, 48, 4), ("stage3", 8, 8, 1)):
    h, w = 2752 // sc, 1856 // sc
    torch.manual_seed(1)
    feats = [torch.randn(C, h, w, device="cuda") for _ in range(5)]
    p34 = ops.compose_projections(torch.from_numpy(s["proj_matrices"][stage]).cuda())
    dv = torch.from_numpy(s["depth_values"]).cuda()
    lo, hi = float(dv[0]), float(dv[-1])
    base = torch.full((h, w), 0.5 * (lo + hi), device="cuda") + 3.0 * torch.randn(h, w, device="cuda")
    depth = ops.depth_range_samples(base, D, (hi - lo) / 384 * sc) if stage != "stage1" else ops.depth_range_samples(dv, D, 0.0)
    planar = ops.variance_volume(feats, p34, depth)
    want = planar.to(ops.h16_dtype()).permute(1, 2, 3, 0).contiguous()
    got = ops.variance_volume_cl(feats, p34, depth)
    bad = (got.view(torch.int16) != want.view(torch.int16))
    nb = int(bad.sum())
    print(stage, "mismatching values:", nb, "non-finite in CL:", int((~torch.isfinite(got.float())).sum()), flush=True)
    if nb:
        idx = bad.nonzero()
        print("  d range", int(idx[:, 0].min()), int(idx[:, 0].max()), "y range", int(idx[:, 1].min()), int(idx[:, 1].max()),
              "x range", int(idx[:, 2].min()), int(idx[:, 2].max()), "channels", sorted(set(idx[:, 3].tolist()))[:16])
        print("  first", idx[:12].tolist())
        ys = idx[:, 1].unique(); xs = idx[:, 2].unique()
        print("  distinct y", ys[:20].tolist(), "distinct x", xs[:40].tolist())
