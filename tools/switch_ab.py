"""ms per reference view of one model (h16 mode, 2752 x 1856, 5 views) with kernels / host forms of config.KERNELS switched off one
set at a time:  python tools/switch_ab.py adamvs "" slice_graph hand_over  """
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import config, ops, predict, synthetic as S

name = sys.argv[1]
sets = sys.argv[2:] or [""]
net = predict.build_model(name, 384)
S.fill_state_dict_(net.state_dict(), 1)
net = net.cuda().eval()
s = predict.SyntheticBlock(1, 5, 2752, 1856, 384)[0]
imgs = torch.from_numpy(s["imgs"])[None].cuda()
pm = {k: torch.from_numpy(v)[None].cuda() for k, v in s["proj_matrices"].items()}
dv = torch.from_numpy(s["depth_values"])[None].cuda()
ops.set_conv_precision("h16")
for rnd in range(2):
    for off in sets:
        config.switches["D3D_KERNELS_OFF"] = off
        with torch.no_grad():
            for _ in range(2):
                net(imgs, pm, dv)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                net(imgs, pm, dv)
            torch.cuda.synchronize()
        print("%-10s off=%-32r %.2f ms per view" % (name, off, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
