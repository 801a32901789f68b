"""conv0 of a feature trunk at full resolution: the fused launch (d3d_conv2d_k3_pair3_bf16x3) against the two launches.
    python tools/conv0_pair_bench.py [H W]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2752, 1856)
x = torch.rand(3, H, W, device="cuda")
w0, w1 = 0.4 * torch.randn(8, 3, 3, 3, device="cuda"), 0.2 * torch.randn(8, 8, 3, 3, device="cuda")
s0, t0, s1, t1 = [torch.randn(8, device="cuda") for _ in range(4)]
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with ops.fp32_convs():
    a = timeit(lambda: ops.conv2d_k3_pair3(x, w0, s0, t0, 1, w1, s1, t1, 1))
    b = timeit(lambda: ops.conv2d_k3(x, w0, s0, t0, None, act=1))
    mid = ops.conv2d_k3(x, w0, s0, t0, None, act=1)
    c = timeit(lambda: ops.conv2d_k3(mid, w1, s1, t1, None, act=1))
mb = (3 + 8) * H * W * 4 / 1e6
print("%d x %d: fused %.1f us (%.0f MB = %.2f of 8 TB/s) | 3 -> 8 %.1f us + 8 -> 8 %.1f us = %.1f us" % (H, W, a, mb, mb / a / 8.0, b, c, b + c))
