"""fp32 mode of the stride-1 C_out <= 16 layers of the CostRegNets at the cascade shapes: the split-operand matrix-core kernel
(d3d_conv3d_k3_zs_bf16x3) against the kernels it replaces (D3D_CONV_C8X3=0).   python tools/x3_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import config, ops
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for tag, Ci, Co, D, h, w in (("conv0 stage1", 32, 8, 48, 464, 688), ("conv0 stage2", 16, 8, 32, 928, 1376), ("conv0 stage3", 8, 8, 8, 1856, 2752),
                             ("prob stage1", 8, 1, 48, 464, 688), ("prob stage2", 8, 1, 32, 928, 1376), ("prob stage3", 8, 1, 8, 1856, 2752),
                             ("conv2 stage1", 16, 16, 24, 232, 344), ("conv2 stage2", 16, 16, 16, 464, 688), ("conv2 stage3", 16, 16, 4, 928, 1376)):
    x = torch.randn(Ci, D, h, w, device="cuda"); wt = torch.randn(Co, Ci, 3, 3, 3, device="cuda") * 0.1
    r = []
    for sw in ("all", "0"):
        config.switches["D3D_CONV_C8X3"] = sw
        r.append(timeit(lambda: ops.conv3d_k3(x, wt, relu=True)))
    config.switches["D3D_CONV_C8X3"] = "1"
    print("%s %2d -> %2d at %2d x %4d x %4d: split operands %.3f ms, previous kernel %.3f ms" % (tag, Ci, Co, D, h, w, r[0], r[1]), flush=True)
for tag, Co, D, h, w in (("conv11 stage1", 8, 24, 232, 344), ("conv11 stage2", 8, 16, 464, 688), ("conv11 stage3", 8, 4, 928, 1376)):
    x = torch.randn(16, D, h, w, device="cuda"); wt = torch.randn(16, Co, 3, 3, 3, device="cuda") * 0.1
    sk = torch.randn(Co, 2 * D, 2 * h, 2 * w, device="cuda")
    r = []
    for sw in ("1", "0"):
        config.switches["D3D_CONV_C8X3"] = sw
        r.append(timeit(lambda: ops.convtranspose3d_k3s2(x, wt, skip=sk, relu=True)))
    config.switches["D3D_CONV_C8X3"] = "1"
    print("%s 16 -> %2d (transposed, x2) from %2d x %4d x %4d: split operands %.3f ms, previous kernel %.3f ms" % (tag, Co, D, h, w, r[0], r[1]), flush=True)
