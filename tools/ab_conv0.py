"""A/B of the activation formats of the z-streaming bf16 convolution at the conv0 shapes (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from deep3d_aerial_amd import ops
from conv_bench import timeit
ops.set_conv_precision("h16")
shapes = [(32, 48, 464, 688), (16, 32, 928, 1376), (8, 8, 1856, 2752)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if s[0] == int(sys.argv[1])]
for (Ci, D, h, w) in shapes:
    x = torch.randn(Ci, D, h, w, device="cuda")
    xc = ops.to_cl(x)
    wt = torch.randn(8, Ci, 3, 3, 3, device="cuda") * 0.1
    for name, fn in [("planar->planar", lambda: ops.conv3d_k3_cl(x, wt, out_cl=False)), ("planar->CL", lambda: ops.conv3d_k3_cl(x, wt, out_cl=True)),
                     ("CL->CL", lambda: ops.conv3d_k3_cl(xc, wt, out_cl=True)), ("CL->planar", lambda: ops.conv3d_k3_cl(xc, wt, out_cl=False))]:
        print(Ci, name, "%.3f ms" % timeit(fn, 3), flush=True)
