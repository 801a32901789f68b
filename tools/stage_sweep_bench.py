"""Cost-volume kernels at the three cascade stage shapes (config 3), tiled vs direct path."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops, synthetic as S

H, W = 1856, 2752
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
        depth = torch.stack([base + (d - D / 2) * float(dv[1] - dv[0]) / 384 * sc for d in range(D)]).contiguous()
    else:
        depth = torch.from_numpy(S.uniform_depths(dv, D)).cuda()
    vw = torch.rand(4, h, w, device="cuda")
    for path in (sys.argv[1:] or ("tiled", "direct")):
        os.environ["D3D_FORCE_PATH"] = path
        try:
            t1 = timeit(lambda: ops.variance_volume(feats, p34, depth))
            t2 = timeit(lambda: ops.weighted_corr(feats, p34, vw, depth))
            print("%s C=%d D=%d %dx%d %-6s variance %.3f ms  weighted %.3f ms  (%.1f Gvoxel/s)" % (
                tag, C, D, h, w, path, t1, t2, D * h * w / t1 / 1e6), flush=True)
        except RuntimeError as e:
            print(tag, path, "unsupported:", str(e)[:80])
