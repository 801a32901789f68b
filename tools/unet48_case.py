"""The 48-channel layers of AdaMVS's pair-visibility UNet (adamvs.py:198-238) in isolation: kernels launched and device time per
layer kind and level (GPU box).  argv[1]: fp32 (default) | bf16."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops
ops.set_conv_precision(sys.argv[1] if len(sys.argv) > 1 else "fp32")
from torch.profiler import profile, ProfilerActivity
w = torch.randn(48, 48, 3, 3, device="cuda") * 0.05
s = torch.ones(48, device="cuda"); t = torch.zeros(48, device="cuda")
for (H, W) in [(464, 688), (232, 344), (116, 172), (58, 86)]:
    x = torch.randn(48, H, W, device="cuda")
    for kind in ("s1", "s2", "t2"):
        if kind == "t2" and H == 464:
            continue
        f = {"s1": lambda: ops.conv2d_k3(x, w, s, t, None, act=1), "s2": lambda: ops.conv2d_k3(x, w, s, t, None, act=1, stride=2),
             "t2": lambda: ops.convtranspose2d_k3s2(x, w, s, t, None, act=1)}[kind]
        for _ in range(3): f()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(4): f()
            torch.cuda.synchronize()
        ev = [(e.key[:70], e.count // 4, e.self_device_time_total / 4) for e in prof.key_averages() if e.self_device_time_total > 0]
        tot = sum(e[2] for e in ev)
        print("%s %dx%d: %.1f us per call: %s" % (kind, H, W, tot, "; ".join("%s x%d %.0fus" % e for e in sorted(ev, key=lambda e: -e[2])[:4])))
