"""RED-Net at 2752 x 1856, 5 views, h16 mode: ms per cascade stage (HIP events around InferDepthNet.forward) and for the feature
network + the rest, with the slice loop as launches / as a captured graph and the four-stream slice on / off:
    python tools/red_stage_times.py ["" red_graph red_streams red_graph,red_streams]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import config, ops, predict, synthetic as S
from deep3d_aerial_amd import msrednet

sets = sys.argv[1:] or ["", "red_graph", "red_streams", "red_graph,red_streams"]
net = predict.build_model("msrednet", 384)
S.fill_state_dict_(net.state_dict(), 1)
net = net.cuda().eval()
s = predict.SyntheticBlock(1, 5, 2752, 1856, 384)[0]
imgs = torch.from_numpy(s["imgs"])[None].cuda()
pm = {k: torch.from_numpy(v)[None].cuda() for k, v in s["proj_matrices"].items()}
dv = torch.from_numpy(s["depth_values"])[None].cuda()
ops.set_conv_precision("h16")
marks = []
inner = msrednet.InferDepthNet.forward
def timed(self, *a, **k):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); out = inner(self, *a, **k); e1.record()
    marks.append((e0, e1))
    return out
msrednet.InferDepthNet.forward = timed
for off in sets:
    config.switches["D3D_KERNELS_OFF"] = off
    with torch.no_grad():
        for _ in range(3):
            net(imgs, pm, dv)
        torch.cuda.synchronize()
        marks.clear()
        n = 5
        t0 = time.perf_counter()
        for _ in range(n):
            net(imgs, pm, dv)
        torch.cuda.synchronize()
        view = (time.perf_counter() - t0) / n * 1e3
    st = [sum(a.elapsed_time(b) for a, b in marks[i::3]) / n for i in range(3)]
    print("off=%-28r view %.2f ms: stage1 %.2f  stage2 %.2f  stage3 %.2f  features+rest %.2f" % (off, view, st[0], st[1], st[2], view - sum(st)), flush=True)
