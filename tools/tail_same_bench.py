"""d3d_slice_tail_regress_same_h16 against the two launches it replaces, at the last cascade stage's shape (state2 16 x 1376 x 928) and
at RED-Net's three stages:  python tools/tail_same_bench.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
def timed(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
rng = np.random.default_rng(1)
for (h, w) in [(1376, 928), (688, 464), (344, 232)]:
    s2, s1 = dev(rng.standard_normal((16, h, w))), dev(rng.standard_normal((8, 2 * h, 2 * w)))
    wu, bu = dev(0.2 * rng.standard_normal((16, 8, 3, 3))), dev(rng.standard_normal(8))
    wh, bh = dev(0.3 * rng.standard_normal((1, 8, 3, 3))), dev(rng.standard_normal(1))
    dpl = dev(600 + 50 * rng.standard_normal((2 * h, 2 * w)))
    acc = [torch.zeros((2 * h, 2 * w), device="cuda") for _ in range(3)]
    with ops.h16_convs():
        t1 = timed(lambda: ops.slice_tail_regress_same(s2, wu, bu, s1, False, wh, bh, dpl, *acc))
        def two():
            up = ops.convtranspose2d_k3s2(s2, wu, None, bu, s1, skip_after_act=False, act=1)
            ops.slice_head_regress(up, wh, bh, False, dpl, *acc)
        t2 = timed(two)
    mb = (16 * h * w + 8 * 4 * h * w + 7 * 4 * h * w) * 4 / 1e6   # state2 + state1 + dplane + 3 maps read and written
    print("state2 16 x %d x %d: fused %.1f us (%.2f TB/s on %.0f MB), two launches %.1f us" % (h, w, t1, mb / t1, mb, t2), flush=True)
