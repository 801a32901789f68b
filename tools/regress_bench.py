"""HBM-roofline check of the streaming regression kernels (a7, a9, a10) at the cascade-stage shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops

H, W = 1856, 2752


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def line(tag, ms, nbytes):
    print("%-44s %8.3f ms  %7.1f GB/s algorithmic  (%4.1f %% of 8 TB/s)" % (tag, ms, nbytes / ms / 1e6, 100 * nbytes / (ms * 1e-3) / 8e12), flush=True)


for (tag, D, sc) in [("stage1", 48, 4), ("stage2", 32, 2), ("stage3", 8, 1)]:
    h, w = H // sc, W // sc
    cost = torch.randn(D, h, w, device="cuda")
    depth = torch.rand(D, h, w, device="cuda") * 100 + 500
    line("softargmin_conf4 %s [%d,%d,%d] per-pixel depth" % (tag, D, h, w), timeit(lambda: ops.softargmin_conf4(cost, depth)),
         4.0 * (2 * D + 2) * h * w)
    dvec = torch.linspace(400, 800, D, device="cuda")
    line("softargmin_conf4 %s [%d,%d,%d] plane depth" % (tag, D, h, w), timeit(lambda: ops.softargmin_conf4(cost, dvec)),
         4.0 * (D + 2) * h * w)
    cur = torch.rand(h, w, device="cuda") * 100 + 500
    line("depth_range_samples %s -> [%d,%d,%d]" % (tag, D, h, w), timeit(lambda: ops.depth_range_samples(cur, D, 1.5)),
         4.0 * (D + 1) * h * w)
    reg = torch.randn(h, w, device="cuda")
    mp, sd, sp = (torch.zeros(h, w, device="cuda") for _ in range(3))
    line("online_regress_update %s [%d,%d]" % (tag, h, w), timeit(lambda: ops.online_regress_update(reg, cur, mp, sd, sp)),
         4.0 * 8 * h * w)
    if sc > 1:
        x = torch.randn(D, h, w, device="cuda")
        line("resize_bilinear %s [%d,%d,%d] -> x2" % (tag, D, h, w), timeit(lambda: ops.resize_bilinear(x, 2 * h, 2 * w)),
             4.0 * D * h * w * 5)
