"""conv11 + prob of the three CostRegNets at the cascade's full-size shapes: the fused kernel (d3d_convtranspose3d_prob_cl_h16)
next to the two launches it replaces (x-folded transposed layer, k_z-folded probability layer)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops  # noqa: E402
from conv_bench import timeit  # noqa: E402

SHAPES = [("stage1", 24, 232, 344), ("stage2", 16, 464, 688), ("stage3", 4, 928, 1376)]   # coarse [D, H, W] (W = the long side)
for tag, D, H, W in SHAPES:
    x = torch.randn(D, H, W, 16, device="cuda").to(ops.h16_dtype())
    sk = torch.randn(2 * D, 2 * H, 2 * W, 8, device="cuda").to(ops.h16_dtype())
    wt = torch.randn(16, 8, 3, 3, 3, device="cuda") * 0.1
    wp = torch.randn(1, 8, 3, 3, 3, device="cuda") * 0.1
    sc, sh, bp = torch.rand(8, device="cuda") + 0.5, torch.randn(8, device="cuda"), torch.randn(1, device="cuda")
    vox = 8 * D * H * W

    def two():
        y = ops.convtranspose3d_k3s2_cl(x, wt, sc, sh, sk, relu=True)
        return ops.conv3d_k3_cl(y, wp, None, bp, None, relu=False, stride=1, out_cl=False)

    def t2():
        return ops.convtranspose3d_k3s2_cl(x, wt, sc, sh, sk, relu=True)

    def one():
        return ops.convtranspose3d_prob_cl(x, wt, sc, sh, sk, wp, bp)

    if not os.environ.get("T2P_NOCHECK"):
        assert torch.equal(one(), two()[0])
    a, a1, b = timeit(two, 5), timeit(t2, 5), timeit(one, 5)
    print("%s %2d x %4d x %4d coarse | two launches %.3f ms (conv11 %.3f, 56 B/voxel: %.0f GB/s) | fused %.3f ms (24 B/voxel: %.0f GB/s)"
          % (tag, D, H, W, a, a1, 56 * vox / a / 1e6, b, 24 * vox / b / 1e6), flush=True)
