"""Does the slice loop capture into a HIP graph on this stack?  One AdaMVS view at a small size, serial loop vs captured loop with
1 and 3 chains; run under `timeout`.  python tools/graph_probe.py [chains ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import adamvs, config, ops, predict, synthetic as S

H, W = int(os.environ.get("PROBE_H", 256)), int(os.environ.get("PROBE_W", 384))
net = predict.build_model("adamvs", 384)
S.fill_state_dict_(net.state_dict(), 1)
net = net.cuda().eval()
imgs, pm, dv = S.model_inputs(5, H, W, 384, 5)
args = (torch.from_numpy(imgs).cuda(), {k: torch.from_numpy(v).cuda() for k, v in pm.items()}, torch.from_numpy(dv).cuda())
ops.set_conv_precision("h16")
config.switches["D3D_KERNELS_OFF"] = "slice_graph"
with torch.no_grad():
    ref = net(*args)["depth"].clone()
torch.cuda.synchronize()
print("serial ok", flush=True)
adamvs.SliceLoopGraph.UP_ONLY = os.environ.get("PROBE_UP_ONLY", "0") == "1"
for chains in [int(x) for x in (sys.argv[1:] or ["1", "3"])]:
    adamvs.SliceLoopGraph.CHAINS = chains
    adamvs.SliceLoopGraph._cache.clear()
    config.switches["D3D_KERNELS_OFF"] = ""
    with torch.no_grad():
        for i in range(3):
            print("chains %d call %d ..." % (chains, i), flush=True)
            out = net(*args)["depth"]
            torch.cuda.synchronize()
            print("   equal to serial:", bool(torch.equal(out, ref)), flush=True)
        t0 = time.perf_counter()
        for i in range(5):
            net(*args)
        torch.cuda.synchronize()
        print("chains %d: %.2f ms per view" % (chains, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
