"""BASELINE config 5 shape (7 views x 512 planes, features 32 x 928 x 688, fp32 here): tiled vs direct path."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops, synthetic as S

V, C, D, h, w = 7, 32, 512, 928, 688
proj, dv = S.make_scene(V, h, w, D, seed=5)
feats = [torch.from_numpy(f).cuda() for f in S.make_features(V, C, h, w, seed=5)]
p34 = ops.compose_projections(torch.from_numpy(proj).cuda())
depth = torch.from_numpy(S.uniform_depths(dv, D)).cuda()
out = torch.empty((C, D, h, w), dtype=torch.float32, device="cuda")
res = {}
for path in ("tiled", "direct"):
    os.environ["D3D_FORCE_PATH"] = path
    ops.variance_volume(feats, p34, depth, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ops.variance_volume(feats, p34, depth, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    res[path] = out[:, ::64].clone()
    print("config 5 (7 views x 512 planes x 32x928x688 fp32) %-6s %8.2f ms  %6.2f Gvoxel/s  %5.1f %% of 8 TB/s" % (
        path, ms, D * h * w / ms / 1e6, 100 * (4.0 * C * D * h * w + 4.0 * C * V * h * w) / (ms * 1e-3) / 8e12))
print("max |tiled - direct| on sampled planes: %.3g" % (res["tiled"] - res["direct"]).abs().max().item())
