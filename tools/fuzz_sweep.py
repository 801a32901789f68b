"""Random-scene cross-check of the ring (tiled) and window cost-volume kernels against the direct-gather kernel (GPU box):
ring vs direct within the gather tolerance, window vs ring bit for bit (the same arithmetic per sample)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import config, ops, synthetic as S

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
n_cases = int(os.environ.get("FUZZ_CASES", "40"))
worst = 0.0
for case in range(n_cases):
    V = int(rng.integers(2, 8))
    C = int(rng.choice([8, 16, 24, 32]))
    h = int(rng.integers(5, 120)); w = int(rng.integers(5, 200))
    D = int(rng.integers(1, 70)) if case % 3 else int(rng.integers(1, 17))   # (every third case shallow: the 32 x 8-pixel patches)
    sweep = float(rng.uniform(0.5, 40.0)); yaw = float(rng.uniform(0.0, 30.0))
    per_pixel = bool(rng.integers(0, 2))
    mode = str(rng.choice(["variance", "weighted", "pair"]))
    proj, dv = S.make_scene(V, h, w, D, sweep_px=sweep, seed=case, yaw_deg=yaw)
    feats = [torch.from_numpy(f).cuda() for f in S.make_features(V, C, h, w, seed=case)]
    p34 = ops.compose_projections(torch.from_numpy(proj).cuda())
    if per_pixel:
        depth = torch.from_numpy(np.sort(rng.uniform(dv[0], dv[1], (D, h, w)).astype(np.float32), 0)).cuda()
    else:
        depth = torch.from_numpy(S.uniform_depths(dv, D)).cuda()
    wts = torch.rand(V - 1, h, w, device="cuda")
    outs = {}
    for path in ("tiled", "direct", "window"):
        config.switches["D3D_FORCE_PATH"] = path
        try:
            if mode == "variance":
                y = ops.variance_volume(feats, p34, depth)
            elif mode == "weighted":
                y = ops.weighted_corr(feats, p34, wts, depth)
            else:
                y = ops.pair_corr_mean(feats[0], feats[1], p34[:1].contiguous(), depth)
        except RuntimeError as e:
            y = None
        outs[path] = y
    torch.cuda.synchronize()
    tag = "%-8s V=%d C=%2d %3dx%3d D=%2d sweep=%4.1f yaw=%4.1f %s" % (mode, V, C, h, w, D, sweep, yaw, "pixel" if per_pixel else "plane")
    if outs["tiled"] is None:
        print("%-66s (tiled path unsupported)" % tag, flush=True)
        continue
    a, b = outs["tiled"], outs["direct"]
    err = (a - b).abs().max().item()
    rel = (a - b).abs().mean().item() / max(b.abs().mean().item(), 1e-9)
    worst = max(worst, rel)
    bad = (not torch.isfinite(a).all().item()) or rel > 5e-5 or err > 2e-3
    cl_note = ""
    if mode == "variance" and C % 8 == 0:   # the channel-last bf16 volume must be the rounding of the ring kernel's planar one
        config.switches["D3D_FORCE_PATH"] = "tiled"
        cl = ops.variance_volume_cl(feats, p34, depth)
        want = a.to(ops.h16_dtype()).permute(1, 2, 3, 0).contiguous()
        nbad = int((cl.view(torch.int16) != want.view(torch.int16)).sum())
        cl8 = ops.variance_volume_cl(feats, p34, depth, layout="cl8")
        nbad += int((ops.cl8_to_cl(cl8).view(torch.int16) != want.view(torch.int16)).sum())
        cl_note = " | channel-last (both layouts): %d differing values" % nbad
        bad = bad or nbad > 0
    if outs["window"] is not None:   # the window kernel (forced: any depth of sweep) repeats the ring kernel's arithmetic exactly
        nw = int((outs["window"] != a).sum())
        cl_note += " | window: %d differing values" % nw
        bad = bad or nw > 0
        if mode == "variance" and C % 8 == 0:
            config.switches["D3D_FORCE_PATH"] = "window"
            clw = ops.variance_volume_cl(feats, p34, depth)
            wantw = outs["window"].to(ops.h16_dtype()).permute(1, 2, 3, 0).contiguous().view(torch.int16)
            nwc = int((clw.view(torch.int16) != wantw).sum())
            nwc += int((ops.cl8_to_cl(ops.variance_volume_cl(feats, p34, depth, layout="cl8")).view(torch.int16) != wantw).sum())
            cl_note += ", its channel-last form %d" % nwc
            bad = bad or nwc > 0
    print("%-66s rel-L1 %.2e max-abs %.2e%s %s" % (tag, rel, err, cl_note, "  <-- MISMATCH" if bad else ""), flush=True)
    if bad:
        worst = 1.0
print("worst rel-L1 %.2e over %d cases" % (worst, n_cases))
sys.exit(0 if worst < 5e-5 else 1)
