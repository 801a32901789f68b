#!/usr/bin/env python3
"""Where the HOST time of one reference view goes: cProfile of a few Infer_* forwards at
2752x1856 x 5 views with the feature pyramids cached (the steady state of a flight strip), beside the wall time of the
forward without a device sync (= the host's share) and with one (= what the GPU needs).
    python tools/host_profile.py [casmvsnet|adamvs|msrednet] [bf16|fp32]"""
import cProfile, io, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import config, predict, synthetic as S

model = sys.argv[1] if len(sys.argv) > 1 else "casmvsnet"
prec = sys.argv[2] if len(sys.argv) > 2 else "h16"
config.switches["D3D_CONV_PRECISION"] = prec
from deep3d_aerial_amd import ops
net = predict.build_model(model, 384)
S.fill_state_dict_(net.state_dict(), 1)
net = net.cuda().eval()
s = predict.SyntheticBlock(1, 5, 2752, 1856, 384)[0]
imgs = torch.from_numpy(s["imgs"])[None].cuda()
pm = {k: torch.from_numpy(v)[None].cuda() for k, v in s["proj_matrices"].items()}
dv = torch.from_numpy(s["depth_values"])[None].cuda()
from deep3d_aerial_amd.dataset import FeatureCache
net.feature_cache = FeatureCache(8 << 30)
keys = ["img%d" % i for i in range(5)]
reps = 6
with torch.no_grad():
    for _ in range(2):
        net(imgs, pm, dv, image_keys=keys)
    torch.cuda.synchronize()
    host = []
    t_all = time.perf_counter()
    for _ in range(reps):
        t0 = time.perf_counter()
        net(imgs, pm, dv, image_keys=keys)
        host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t_all) / reps
    print("%s %s: forward returns after %.2f ms (median, host side incl. the model's one sync); %.2f ms per view with the GPU drained"
          % (model, prec, sorted(host)[reps // 2] * 1e3, wall * 1e3))
    disp = []
    for _ in range(reps):
        torch.cuda.synchronize()          # idle GPU: the forward's own sync returns at once, the rest is queueing
        t0 = time.perf_counter()
        net(imgs, pm, dv, image_keys=keys)
        disp.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print("host dispatch of one view on an idle GPU (Python + launches, no waiting): %.2f ms (median of %d)" % (sorted(disp)[reps // 2] * 1e3, reps))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(reps):
        net(imgs, pm, dv, image_keys=keys)
    pr.disable()
    torch.cuda.synchronize()
out = io.StringIO()
st = pstats.Stats(pr, stream=out)
st.sort_stats("tottime").print_stats(45)
txt = out.getvalue().replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/", "")
print("cProfile over %d forwards (divide by %d):" % (reps, reps))
print(txt[:9000])
