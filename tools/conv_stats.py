"""Per-phase cycle breakdown of conv_stream_kernel (needs the -DD3D_CONV_STATS debug build: tools/run_convstats.sh)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import _lib, ops  # noqa: E402

lib = _lib.load()
lib.d3d_conv_stream_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
NAMES = ["issue", "land", "sweep", "flush", "barrier", "total", "steps", "wgs"]


def report(tag, fn):
    fn()
    lib.d3d_conv_stream_stats(None, 1)
    fn()
    buf = (ctypes.c_ulonglong * 8)()
    lib.d3d_conv_stream_stats(buf, 1)
    v = list(buf)
    wg = max(v[7], 1)
    print("%-22s wgs=%6d chunks/wg=%5.1f | per-WG cycles: " % (tag, v[7], v[6] / wg) +
          "  ".join("%s=%.0f" % (NAMES[i], v[i] / wg) for i in range(6)), flush=True)


H, W = 1856, 2752
x = torch.randn(8, 8, H, W, device="cuda")
w = torch.randn(8, 8, 3, 3, 3, device="cuda") * 0.1
report("s3 conv0 8->8", lambda: ops.conv3d_k3(x, w))
w1 = torch.randn(1, 8, 3, 3, 3, device="cuda") * 0.1
report("s3 prob 8->1", lambda: ops.conv3d_k3(x, w1, relu=False))
w16 = torch.randn(16, 8, 3, 3, 3, device="cuda") * 0.1
report("s3 conv1 8->16 s2", lambda: ops.conv3d_k3(x, w16, stride=2))
del x
x = torch.randn(16, 4, H // 2, W // 2, device="cuda")
wt = torch.randn(16, 8, 3, 3, 3, device="cuda") * 0.1
report("s3 conv11T 16->8", lambda: ops.convtranspose3d_k3s2(x, wt))
w2 = torch.randn(16, 16, 3, 3, 3, device="cuda") * 0.1
report("s3 conv2 16->16", lambda: ops.conv3d_k3(x, w2))
del x
x = torch.randn(32, 48, H // 4, W // 4, device="cuda")
w0 = torch.randn(8, 32, 3, 3, 3, device="cuda") * 0.1
report("s1 conv0 32->8", lambda: ops.conv3d_k3(x, w0))
del x
# 2D slice-regulariser layers at stage-3 resolution (AdaMVS / RED-Net)
x = torch.randn(8, H, W, device="cuda")
h8 = torch.randn(8, H, W, device="cuda")
wg = torch.randn(16, 16, 3, 3, device="cuda") * 0.1
report("2D gates 8+8->16", lambda: ops.conv2d_k3(x, wg, x2=h8))
wc = torch.randn(8, 16, 3, 3, device="cuda") * 0.1
report("2D cand 8+8->8", lambda: ops.conv2d_k3(x, wc, x2=h8))
w1 = torch.randn(8, 8, 3, 3, device="cuda") * 0.1
report("2D conv1 8->8", lambda: ops.conv2d_k3(x, w1, act=1))
wo = torch.randn(1, 8, 3, 3, device="cuda") * 0.1
report("2D out 8->1", lambda: ops.conv2d_k3(x, wo))
