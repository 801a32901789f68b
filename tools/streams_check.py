"""A model at 2752 x 1856 with one of the multi-stream forms (red_streams: RED-Net's conv-GRU levels of a slice; fpn_streams: the feature
pyramids of a view set) against the one-stream forward and against itself -- relative L1 of the stage depths, pixels that differ:
    python tools/streams_check.py msrednet red_streams | casmvsnet fpn_streams | adamvs fpn_streams | adamvs pair_streams [extra forwards]"""
import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from deep3d_aerial_amd import config, predict, synthetic as S, ops
ops.set_conv_precision("h16")
model, switch = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("msrednet", "red_streams")
net = predict.build_model(model, 384); S.fill_state_dict_(net.state_dict(), 1); net = net.cuda().eval()
s = predict.SyntheticBlock(1, 5, 2752, 1856, 384)[0]
imgs = torch.from_numpy(s["imgs"])[None].cuda()
pm = {k: torch.from_numpy(v)[None].cuda() for k, v in s["proj_matrices"].items()}
dv = torch.from_numpy(s["depth_values"])[None].cuda()
outs = []
with torch.no_grad():
    for off in ("", switch, "", ""):
        config.switches["D3D_KERNELS_OFF"] = off
        o = net(imgs, pm, dv); torch.cuda.synchronize()
        outs.append({k: o[k]["depth"].clone() for k in ("stage1", "stage2", "stage3")})
# and a longer run of the multi-stream form against the first forward (a rare ordering hazard would show as a differing pixel)
extra = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bad = 0
with torch.no_grad():
    config.switches["D3D_KERNELS_OFF"] = ""
    for _ in range(extra):
        o = net(imgs, pm, dv); torch.cuda.synchronize()
        bad += sum(int((o[k]["depth"] != outs[0][k]).sum()) for k in ("stage1", "stage2", "stage3"))
if extra:
    print("%d more forwards with every multi-stream form on: %d differing pixels" % (extra, bad))
for k in ("stage1", "stage2", "stage3"):
    a = outs[0][k]
    print(k, [float(((o[k] - a).abs().mean() / a.abs().mean())) for o in outs[1:]], [int((o[k] != a).sum()) for o in outs[1:]])
