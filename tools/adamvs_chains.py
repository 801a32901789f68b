"""AdaMVS view (2752 x 1856, 5 views, h16) with the captured slice loop as 1 / 2 / 3 chains (adamvs.SliceLoopGraph.CHAINS):
    python tools/adamvs_chains.py 3 1 2 3"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import adamvs, ops, predict, synthetic as S

s = predict.SyntheticBlock(1, 5, 2752, 1856, 384)[0]
imgs = torch.from_numpy(s["imgs"])[None].cuda()
pm = {k: torch.from_numpy(v)[None].cuda() for k, v in s["proj_matrices"].items()}
dv = torch.from_numpy(s["depth_values"])[None].cuda()
ops.note_depth_range(dv, s["depth_values"][0], s["depth_values"][-1])
ops.set_conv_precision("h16")
for chains in [int(a) for a in sys.argv[1:]] or [3, 1, 2]:
    adamvs.SliceLoopGraph.CHAINS = chains
    adamvs.SliceLoopGraph._cache.clear()
    net = predict.build_model("adamvs", 384)
    S.fill_state_dict_(net.state_dict(), 1)
    net = net.cuda().eval()
    with torch.no_grad():
        for _ in range(3):
            net(imgs, pm, dv)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            net(imgs, pm, dv)
        torch.cuda.synchronize()
    print("chains %d: %.2f ms per view" % (chains, (time.perf_counter() - t0) / 6 * 1e3), flush=True)
    del net
