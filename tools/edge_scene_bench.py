"""How the shallow sweeps behave when the depth map handed down by the previous stage has DISCONTINUITIES (building edges): the
stage-3 / stage-2 shapes with a base depth that jumps by `jump` hypothesis intervals on a checkerboard of `block`-pixel squares.
Patches that straddle a jump need wide windows (or chunks, or the gather fallback).   PPI=1.0 python tools/edge_scene_bench.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import config, ops, synthetic as S

H, W = 1856, 2752
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (tag, C, D, sc) in (("stage2", 16, 32, 2), ("stage3", 8, 8, 1)):
    h, w = H // sc, W // sc
    ppi = float(os.environ.get("PPI", "1.0"))   # pixels at full resolution per base depth interval (see tools/ppi_sweep_bench.py)
    proj, dv = S.make_scene(5, h, w, 384, sweep_px=ppi * 384 / sc, seed=3)
    feats = [torch.randn(C, h, w, device="cuda") for _ in range(5)]
    p34 = ops.compose_projections(torch.from_numpy(proj).cuda())
    interval = float(dv[1] - dv[0]) / 384 * sc   # the stage's hypothesis spacing (ratio 2 at stage 2, 1 at stage 3)
    yy, xx = torch.meshgrid(torch.arange(h, device="cuda"), torch.arange(w, device="cuda"), indexing="ij")
    for block, jump in ((0, 0), (250, 8), (250, 32), (60, 8), (60, 32), (60, 96)):   # (not multiples of the 32 x 8 patches)
        base = torch.full((h, w), float(dv.mean()), device="cuda")
        if block:
            base = base + (((yy // block) + (xx // block)) % 2).float() * jump * interval
        depth = ops.depth_range_affine(base.contiguous(), D, interval)
        res = []
        for path in ("", "tiled"):
            config.switches["D3D_FORCE_PATH"] = path
            res.append(timeit(lambda: ops.variance_volume_cl(feats, p34, depth, layout="cl8")))
        config.switches["D3D_FORCE_PATH"] = ""
        frac = 0.0 if not block else 1.0 - (1.0 - 32.0 / block) * (1.0 - 8.0 / block)   # share of 32 x 8 patches that straddle a jump
        print("%s checkerboard %3d px, jump %2d intervals (%.0f %% of the patches straddle one): window kernel %.3f ms, ring kernel %.3f ms"
              % (tag, block, jump, 100 * frac, res[0], res[1]), flush=True)
