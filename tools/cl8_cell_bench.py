"""Weighted-correlation sweep and first conv-GRU cell of a slice at the three cascade stages: planar fp32 volume against CL8 16-bit cells."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops, synthetic as S

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

ops.set_conv_precision("h16")
H, W = 2752, 1856
for tag, C, D, sc in (("stage1", 32, 48, 4), ("stage2", 16, 32, 2), ("stage3", 8, 8, 1)):
    h, w = H // sc, W // sc
    proj, dv = S.make_scene(5, h, w, 384, seed=3)
    feats = [torch.randn(C, h, w, device="cuda") for _ in range(5)]
    p34 = ops.compose_projections(torch.from_numpy(proj).cuda())
    if sc == 4:
        depth = torch.from_numpy(S.uniform_depths(dv, D)).cuda()
    else:
        depth = ops.depth_range_affine(torch.full((h, w), float(dv.mean()), device="cuda"), D, float(dv[1] - dv[0]) / 384 * sc)
    vw = torch.rand(4, h, w, device="cuda")
    t_planar = timeit(lambda: ops.weighted_corr(feats, p34, vw, depth, plane_major=True), 5)
    t_cl8 = timeit(lambda: ops.weighted_corr_cl8(feats, p34, vw, depth), 5)
    planar = ops.weighted_corr(feats, p34, vw, depth, plane_major=True)
    cl8 = ops.weighted_corr_cl8(feats, p34, vw, depth)
    rng = np.random.default_rng(1)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    h0 = dev(rng.standard_normal((8, h, w)))
    w1 = dev(rng.standard_normal((8, C, 3, 3)) / (3.0 * np.sqrt(C)))
    wg, bg = dev(rng.standard_normal((16, 16, 3, 3)) / 12.0), dev(rng.standard_normal(16))
    wc, bc = dev(rng.standard_normal((8, 16, 3, 3)) / 12.0), dev(rng.standard_normal(8))
    out = torch.empty_like(h0)
    c_planar = timeit(lambda: ops.gru_cell_conv_fused(planar[D // 2], h0, w1, wg, bg, wc, bc, 1, out=out))
    c_cl8 = timeit(lambda: ops.gru_cell_conv_fused(cl8[D // 2], h0, w1, wg, bg, wc, bc, 1, out=out))
    s1 = dev(rng.standard_normal((8, h, w)))
    h2 = dev(rng.standard_normal((16, h // 2, w // 2)))
    w2 = dev(rng.standard_normal((16, 8, 3, 3)) / 8.5)
    wg2, bg2 = dev(rng.standard_normal((32, 32, 3, 3)) / 17.0), dev(rng.standard_normal(32))
    wc2, bc2 = dev(rng.standard_normal((16, 32, 3, 3)) / 17.0), dev(rng.standard_normal(16))
    out2 = torch.empty_like(h2)
    c2 = timeit(lambda: ops.gru_cell_conv_fused(s1, h2, w2, wg2, bg2, wc2, bc2, 2, out=out2))
    print("%s   stride-2 cell (8 -> 16 at %dx%d): %.1f us" % (tag, h // 2, w // 2, c2 * 1e3), flush=True)
    print("%s C=%d D=%d %dx%d: sweep planar %.3f ms, CL8 %.3f ms | cell planar %.1f us, CL8 %.1f us" % (tag, C, D, h, w, t_planar, t_cl8, c_planar * 1e3, c_cl8 * 1e3), flush=True)
