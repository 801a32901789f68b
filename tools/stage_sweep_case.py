"""One cascade-stage shape of the cost-volume sweeps, a few launches of each product (workload for tools/run_pmc_hbm.sh /
run_pmc_script.sh: the window kernel's instances differ per product, so a run over ONE stage gives per-kernel counters):
    python tools/stage_sweep_case.py stage1|stage2|stage3 [reps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops, synthetic as S

H, W = 1856, 2752
tag = sys.argv[1] if len(sys.argv) > 1 else "stage3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
C, D, sc, perpix = {"stage1": (32, 48, 4, False), "stage2": (16, 32, 2, True), "stage3": (8, 8, 1, True)}[tag]
h, w = H // sc, W // sc
proj, dv = S.make_scene(5, h, w, 384 // (1 if not perpix else 4), seed=3)
feats = [torch.randn(C, h, w, device="cuda") for _ in range(5)]
p34 = ops.compose_projections(torch.from_numpy(proj).cuda())
if perpix:
    depth = ops.depth_range_affine(torch.full((h, w), float(dv.mean()), device="cuda"), D, float(dv[1] - dv[0]) / 384 * sc)
else:
    depth = torch.from_numpy(S.uniform_depths(dv, D)).cuda()
vw = torch.rand(4, h, w, device="cuda")
for _ in range(reps):
    a = ops.variance_volume(feats, p34, depth)
    b = ops.weighted_corr(feats, p34, vw, depth)
    c = ops.variance_volume_cl(feats, p34, depth, layout="cl8")
torch.cuda.synchronize()
reads = 5 * C * h * w * 4 + (2 * h * w * 4 if perpix else 0)
print("%s C=%d D=%d %dx%d: algorithmic MB per launch: planar %.1f, channel-last bf16 %.1f" % (tag, C, D, h, w, (reads + C * D * h * w * 4) / 1e6, (reads + C * D * h * w * 2) / 1e6))
