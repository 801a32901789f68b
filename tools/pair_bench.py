import sys, os, torch
sys.path.insert(0, "/root/repo")
from deep3d_aerial_amd import config, ops, synthetic as S
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
h, w, C, D = 464, 688, 32, 48
for ppi in (0.25, 1.0, 2.0):
    proj, dv = S.make_scene(5, h, w, 384, sweep_px=ppi * 384 / 4, seed=3)
    feats = [torch.randn(C, h, w, device="cuda") for _ in range(2)]
    p34 = ops.compose_projections(torch.from_numpy(proj).cuda())
    depth = torch.from_numpy(S.uniform_depths(dv, D)).cuda()
    r = []
    for path in ("", "tiled", "direct"):
        config.switches["D3D_FORCE_PATH"] = path
        r.append(timeit(lambda: ops.pair_corr_mean(feats[0], feats[1], p34[0].contiguous(), depth)))
    config.switches["D3D_FORCE_PATH"] = ""
    print("pair correlation C=32 D=48 464x688, %.2f px per interval: window %.3f ms, ring %.3f ms, direct %.3f ms" % (ppi, *r), flush=True)
