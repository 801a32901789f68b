"""conv2d_wide (RED-Net's 64- / 128-channel conv-GRU levels) in isolation at the shapes of the three cascade stages: us per launch,
TFLOP/s and GB/s.   python tools/wide_bench.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops
def timed(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
rng = np.random.default_rng(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
with ops.h16_convs():
    for tag, (H, W) in (("stage 3", (688, 464)), ("stage 2", (344, 232)), ("stage 1", (172, 116))):
        for lvl, C, h, w in (("level 3", 32, H, W), ("level 4", 64, H // 2, W // 2)):
            x, s = dev(rng.standard_normal((C, h, w))), dev(rng.standard_normal((C, h, w)))
            for Co in (2 * C, C):
                wt, b = dev(0.05 * rng.standard_normal((Co, 2 * C, 3, 3))), dev(rng.standard_normal(Co))
                gs = ops.GnStats(2 if Co == 2 * C else 1)
                t = timed(lambda: ops.conv2d_k3(x, wt, None, b, None, act=0, stride=1, x2=s, gn=ops.GnStats(2 if Co == 2 * C else 1)))
                fl = 2.0 * Co * 2 * C * 9 * h * w
                by = (2 * C + Co) * h * w * 4
                print("%s %s: %3d+%3d -> %3d at %4d x %4d: %7.1f us  %6.1f TFLOP/s  %6.1f GB/s" % (tag, lvl, C, C, Co, h, w, t, fl / t / 1e6, by / t / 1e3), flush=True)
