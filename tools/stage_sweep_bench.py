"""Cost-volume kernels at the three cascade stage shapes (config 3), tiled vs direct path."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import config, ops, synthetic as S

H, W = 1856, 2752
AFFINE = os.environ.get("STAGE_AFFINE", "1") != "0"   # STAGE_AFFINE=0: per-pixel hypothesis volumes [D,h,w] (rounds 1-2)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (tag, C, D, sc, perpix) in [("stage1", 32, 48, 4, False), ("stage2", 16, 32, 2, True), ("stage3", 8, 8, 1, True)]:
    h, w = H // sc, W // sc
    proj, dv = S.make_scene(5, h, w, 384 // (1 if not perpix else 4), seed=3)
    feats = [torch.randn(C, h, w, device="cuda") for _ in range(5)]
    p34 = ops.compose_projections(torch.from_numpy(proj).cuda())
    if perpix:
        base = torch.full((h, w), float(dv.mean()), device="cuda")
        if AFFINE:   # the hypotheses as (lo, step) maps (ops.AffineDepth): what the cascades pass since round 3
            depth = ops.depth_range_affine(base, D, float(dv[1] - dv[0]) / 384 * sc)
        else:
            depth = torch.stack([base + (d - D / 2) * float(dv[1] - dv[0]) / 384 * sc for d in range(D)]).contiguous()
    else:
        depth = torch.from_numpy(S.uniform_depths(dv, D)).cuda()
    vw = torch.rand(4, h, w, device="cuda")
    roof = {"stage1": None, "stage2": None, "stage3": None}
    reads = 5 * C * h * w * 4 + (0 if not perpix else (2 if AFFINE else D) * h * w * 4)
    gb_var, gb_cl = (reads + C * D * h * w * 4) / 1e9, (reads + C * D * h * w * 2) / 1e9
    for path in (sys.argv[1:] or ("tiled", "direct")):
        config.switches["D3D_FORCE_PATH"] = path
        try:
            t1 = timeit(lambda: ops.variance_volume(feats, p34, depth))
            t2 = timeit(lambda: ops.weighted_corr(feats, p34, vw, depth))
            t3 = timeit(lambda: ops.variance_volume_cl(feats, p34, depth, layout="cl8")) if path != "direct" else float("nan")
            print("%s C=%d D=%d %dx%d %-6s variance %.3f ms (%.3f of 8 TB/s)  weighted %.3f ms  channel-last bf16 %.3f ms (%.3f)  (%.1f Gvoxel/s)" % (
                tag, C, D, h, w, path, t1, gb_var / t1 / 8.0, t2, t3, gb_cl / t3 / 8.0, D * h * w / t1 / 1e6), flush=True)
        except RuntimeError as e:
            print(tag, path, "unsupported:", str(e)[:80])
