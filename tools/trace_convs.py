"""Counts, per (entry point, input shape, weight shape, stride), the convolution calls of one AdaMVS forward (GPU box): shows
which layers the tile kernels of round 2 take and which stay on round 1's kernels.  TRACE_PRECISION=bf16|fp32 (default bf16),
TRACE_ALL=1 lists every entry point, not just round 1's."""
import os, sys, collections, torch
sys.path.insert(0, os.getcwd())
from deep3d_aerial_amd import predict, ops, synthetic as S
cnt = collections.Counter(); ms = collections.Counter()
timed = os.environ.get("TRACE_TIME") == "1"   # TRACE_TIME=1: inclusive device time per entry point (nested entries count twice)
def wrap(name):
    f = getattr(ops, name)
    def g(x, weight, *a, **k):
        if timed:
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        y = f(x, weight, *a, **k)
        key = (name, tuple(x.shape), tuple(weight.shape), k.get("stride", 1), k.get("act", None), "x2" if k.get("x2") is not None else "", y is not None)
        cnt[key] += 1
        if timed:
            e1.record(); torch.cuda.synchronize(); ms[key] += e0.elapsed_time(e1)
        return y
    setattr(ops, name, g)
for n in ("conv_k3_mfma", "convtranspose_k3s2_mfma", "conv_fold", "conv2d_stream", "conv2d_zs", "conv2d_s2_zs", "convtranspose2d_zs", "conv2d_same", "conv1x1_upskip", "conv2d_wide"):
    if hasattr(ops, n): wrap(n)
net = predict.build_model(os.environ.get("TRACE_MODEL", "adamvs"), 384); S.fill_state_dict_(net.state_dict(), 1); net = net.cuda().eval()
s = predict.SyntheticBlock(1, 5, 2752, 1856, 384)[0]
imgs = torch.from_numpy(s["imgs"])[None].cuda(); pm = {k: torch.from_numpy(v)[None].cuda() for k, v in s["proj_matrices"].items()}; dv = torch.from_numpy(s["depth_values"])[None].cuda()
ops.set_conv_precision(os.environ.get("TRACE_PRECISION", "h16"))
with torch.no_grad():
    if timed:
        net(imgs, pm, dv); cnt.clear(); ms.clear()   # warm-up: packed weights, kernel attributes
    net(imgs, pm, dv)
if timed:
    for k, v in sorted(ms.items(), key=lambda kv: -kv[1])[:40]:
        print("%8.3f ms  x%3d  %s" % (v, cnt[k], k))
    sys.exit(0)
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]):
    if os.environ.get("TRACE_ALL") == "1" or (k[0] in ("conv_k3_mfma", "conv_fold", "convtranspose_k3s2_mfma") and k[-1]):
        print(v, k)
