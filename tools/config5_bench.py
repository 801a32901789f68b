"""BASELINE config 5 shape (7 views x 512 planes, features 32 x 928 x 688) in its own storage type (fp16 features and
cost volume, fp32 arithmetic) and in fp32: LDS-ring kernel vs direct-gather kernel, with the HBM-roofline fraction
(algorithmic bytes = volume written once + V feature maps read once, in the storage type).

    python tools/config5_bench.py [f16|f32|both]      (D3D_FORCE_PATH must be unset: each path runs in a child process)
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

V, C, D, h, w = 7, 32, 512, 928, 688


def run(dtype, path):
    import torch
    from deep3d_aerial_amd import ops, synthetic as S

    proj, dv = S.make_scene(V, h, w, D, seed=5)
    td = torch.float16 if dtype == "f16" else torch.float32
    feats = [torch.from_numpy(f).cuda().to(td) for f in S.make_features(V, C, h, w, seed=5)]
    p34 = ops.compose_projections(torch.from_numpy(proj).cuda())
    depth = torch.from_numpy(S.uniform_depths(dv, D)).cuda()
    out = torch.empty((C, D, h, w), dtype=td, device="cuda")
    ops.variance_volume(feats, p34, depth, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 3
    for _ in range(n):
        ops.variance_volume(feats, p34, depth, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    eb = 2.0 if dtype == "f16" else 4.0
    alg = eb * C * D * h * w + eb * C * V * h * w
    print("config 5 (7 views x 512 planes x 32x928x688) %s %-6s %8.2f ms  %6.2f Gvoxel/s  %.2f GB algorithmic  %.3f of 8 TB/s" % (
        dtype, path, ms, D * h * w / ms / 1e6, alg / 1e9, alg / (ms * 1e-3) / 8e12), flush=True)
    torch.save(out[:, ::64, ::8, ::8].float().cpu(), "/tmp/c5_%s_%s.pt" % (dtype, path))


if __name__ == "__main__":
    if len(sys.argv) > 2:
        run(sys.argv[1], sys.argv[2])
        sys.exit(0)
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    import torch
    for dtype in (["f16", "f32"] if which == "both" else [which]):
        for path in ("tiled", "direct"):
            env = dict(os.environ, D3D_FORCE_PATH=path)
            subprocess.run([sys.executable, os.path.abspath(__file__), dtype, path], env=env, check=True)
        a, b = torch.load("/tmp/c5_%s_tiled.pt" % dtype), torch.load("/tmp/c5_%s_direct.pt" % dtype)
        print("  %s: max |tiled - direct| on sampled voxels: %.3g (max |value| %.3g)" % (dtype, (a - b).abs().max().item(), b.abs().max().item()))
