"""A model's view at 2752 x 1856, 5 views, h16 mode, split into its cascade stages (HIP events around every DepthNet forward) and
the feature pyramids + the rest, per set of config.KERNELS switched off:
    python tools/stage_times.py msrednet "" red_graph red_streams        (default model msrednet, default sets below)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import config, ops, predict, synthetic as S

args = sys.argv[1:]
name = args.pop(0) if args and args[0] in ("casmvsnet", "adamvs", "msrednet", "ucsnet") else "msrednet"
sets = args or ([""] if name != "msrednet" else ["", "red_graph", "red_streams", "red_graph,red_streams"])
net = predict.build_model(name, 384)
S.fill_state_dict_(net.state_dict(), 1)
net = net.cuda().eval()
s = predict.SyntheticBlock(1, 5, 2752, 1856, 384)[0]
imgs = torch.from_numpy(s["imgs"])[None].cuda()
pm = {k: torch.from_numpy(v)[None].cuda() for k, v in s["proj_matrices"].items()}
dv = torch.from_numpy(s["depth_values"])[None].cuda()
if os.environ.get("STAGE_NO_NOTE") != "1":   # as predict.predict_views hands the range over (no host sync per view); STAGE_NO_NOTE=1: the forward reads it back
    ops.note_depth_range(dv, s["depth_values"][0], s["depth_values"][-1])
ops.set_conv_precision("h16")
marks, open_ = [], []
mods = list(net.DepthNet) if isinstance(net.DepthNet, torch.nn.ModuleList) else [net.DepthNet]
def pre(m, a):
    e = torch.cuda.Event(enable_timing=True); e.record(); open_.append(e)
def post(m, a, out):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((open_.pop(), e))
for m in mods:
    m.register_forward_pre_hook(pre); m.register_forward_hook(post)
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
    k = len(marks) // n
    st = [sum(a.elapsed_time(b) for a, b in marks[i::k]) / n for i in range(k)]
    print("%-9s off=%-28r view %.2f ms: %s  features+rest %.2f" % (name, off, view, "  ".join("stage%d %.2f" % (i + 1, t) for i, t in enumerate(st)), view - sum(st)), flush=True)
