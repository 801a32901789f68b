"""Per-layer times of the three CostRegNets of a CasMVSNet view in bf16 mode (channel-last volumes, CL8 input as the sweep kernels
leave it): the layer sequence of CostRegNet.forward_one with an event pair around every launch.  For A/B runs of one kernel
source through tools/kvariants.sh (D3D_LIBRARY)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops, synthetic as S  # noqa: E402
from deep3d_aerial_amd.cas_mvsnet import CostRegNet  # noqa: E402
from deep3d_aerial_amd.module import folded_bn  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ops.set_conv_precision("h16")
total = {}
for tag, C, D, h, w in (("stage1", 32, 48, 464, 688), ("stage2", 16, 32, 928, 1376), ("stage3", 8, 8, 1856, 2752)):
    net = CostRegNet(C).cuda().eval()
    S.fill_state_dict_(net.state_dict(), 3)
    vol = ops.cl_to_cl8(torch.randn(D, h, w, C, device="cuda").to(ops.h16_dtype()))
    times = {}

    def run(name, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        times.setdefault(name, []).append(e0.elapsed_time(e1))
        return out

    with torch.no_grad():
        for rep in range(REPS + 1):
            if rep == 1:
                times.clear()
            c0 = run("conv0", lambda: net.conv0.forward_cl(vol))
            c1 = run("conv1 s2", lambda: net.conv1.forward_cl(c0))
            c2 = run("conv2", lambda: net.conv2.forward_cl(c1))
            c3 = run("conv3 s2", lambda: net.conv3.forward_cl(c2))
            c4 = run("conv4", lambda: net.conv4.forward_cl(c3))
            c5 = run("conv5 s2", lambda: net.conv5.forward_cl(c4))
            y = run("conv6", lambda: net.conv6.forward_cl(c5))
            y = run("conv7 T", lambda: net.conv7.forward_cl(y, c4))
            y = run("conv9 T", lambda: net.conv9.forward_cl(y, c2))
            s11, t11 = folded_bn(net.conv11[1])
            run("conv11+prob", lambda: ops.convtranspose3d_prob_cl(y, net.conv11[0].weight, s11, t11, c0, net.prob.weight, net.prob.bias))
    line = []
    for k, v in times.items():
        ms = sum(v) / len(v)
        total[k] = total.get(k, 0.0) + ms
        line.append("%s %.3f" % (k, ms))
    print("%s | %s | sum %.3f" % (tag, " | ".join(line), sum(sum(v) / len(v) for v in times.values())), flush=True)
    del net, vol
    torch.cuda.empty_cache()
print("view   | " + " | ".join("%s %.3f" % kv for kv in total.items()) + " | sum %.3f" % sum(total.values()))
