"""Cascade-stage sweeps as a function of the scene's DISPARITY PER DEPTH INTERVAL (pixels at full resolution that the farthest
source view moves per base interval; tools/stage_sweep_bench.py's stage 1 is 2.0, its stages 2 / 3 0.125, a model_bench view 0.25):
one camera set for all three stages, channel-last bf16 variance volume (CL8), window kernel (dispatcher) vs ring kernel.
    python tools/ppi_sweep_bench.py [ppi ...]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import config, ops, synthetic as S

H, W = 1856, 2752
def timeit(fn, n=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for ppi in [float(a) for a in (sys.argv[1:] or ("0.25", "0.5", "1", "2"))]:
    line = "ppi %.2f:" % ppi
    for (tag, C, D, sc, ratio) in (("stage1", 32, 48, 4, 8), ("stage2", 16, 32, 2, 2), ("stage3", 8, 8, 1, 1)):
        h, w = H // sc, W // sc
        proj, dv = S.make_scene(5, h, w, 384, sweep_px=ppi * 384 / sc, seed=3)
        feats = [torch.randn(C, h, w, device="cuda") for _ in range(5)]
        p34 = ops.compose_projections(torch.from_numpy(proj).cuda())
        interval = float(dv[1] - dv[0]) / 384
        if tag == "stage1":
            depth = torch.from_numpy(S.uniform_depths(dv, D)).cuda()
        else:
            depth = ops.depth_range_affine(torch.full((h, w), float(dv.mean()), device="cuda"), D, ratio * interval)
        res = []
        for path in ("", "tiled"):
            config.switches["D3D_FORCE_PATH"] = path
            res.append(timeit(lambda: ops.variance_volume_cl(feats, p34, depth, layout="cl8")))
        config.switches["D3D_FORCE_PATH"] = ""
        line += "  %s %.2f px/plane: %.2f ms (ring %.2f)" % (tag, ppi * ratio / sc, res[0], res[1])
        del feats
    print(line, flush=True)
