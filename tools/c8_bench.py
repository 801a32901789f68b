"""conv0 of the three CostRegNets (C_in -> 8, stride 1) at the cascade's full-size shapes: the bf16 matrix-core
z-streaming kernel (d3d_conv3d_k3_c8_h16) next to the round-1 bf16 stream kernel and the fp32 vector-unit kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import config, ops  # noqa: E402
from conv_bench import timeit  # noqa: E402

SHAPES = [("stage1 32->8", 32, 48, 688, 464), ("stage2 16->8", 16, 32, 1376, 928), ("stage3  8->8", 8, 8, 2752, 1856)]
for tag, Ci, D, h, w in SHAPES:
    x = torch.randn(Ci, D, h, w, device="cuda")
    wt = torch.randn(8, Ci, 3, 3, 3, device="cuda") * 0.1
    sc, sh = torch.rand(8, device="cuda") + 0.5, torch.randn(8, device="cuda")
    fn = lambda: ops.conv3d_k3(x, wt, sc, sh, relu=True)
    gb = 4 * (Ci + 8) * D * h * w / 1e9
    res = []
    # (the switches are read from the environment once, at import: a tool changes the table -- config.override -- not os.environ;
    #  the third column is the fp32 mode's kernel: the split-operand matrix-core layer since round 3, the vector-unit one with
    #  D3D_CONV_C8X3=0)
    for mode, env in (("bf16 c8 mfma", {}), ("bf16 stream (r1)", {"D3D_KERNELS_OFF": "c8"}),
                      ("fp32 x3 mfma", None), ("fp32 co8 valu", {"D3D_CONV_C8X3": "0"})):
        ops.set_conv_precision("h16" if mode.startswith("bf16") else "fp32")
        with config.override(**(env or {})):
            ms = timeit(fn, 5)
        res.append("%s %7.3f ms (%5.0f GB/s, %4.2f of 8 TB/s)" % (mode, ms, gb / ms * 1e3, gb / ms / 8.0))
    ops.set_conv_precision(None)
    print("%-14s %2d x %4d x %4d  %.2f GB in+out | %s" % (tag, D, h, w, gb, " | ".join(res)), flush=True)
