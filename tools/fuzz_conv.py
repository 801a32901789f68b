"""Random-shape cross-check of the matrix-core convolutions against the direct VALU kernels (GPU box)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import config, ops

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
n_cases = int(os.environ.get("FUZZ_CASES", "60"))
worst = 0.0
for case in range(n_cases):
    three_d = bool(rng.integers(0, 2))
    Ci = int(rng.choice([1, 3, 4, 8, 12, 16, 24, 32, 48, 64]))
    Co = int(rng.choice([1, 2, 4, 8, 16, 24, 32, 64]))
    D = int(rng.integers(1, 7)) if three_d else 1
    H = int(rng.integers(1, 40))
    W = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 17, 20, 31, 32, 64, 65, 100, 128, 130]))
    stride = int(rng.choice([1, 2]))
    transposed = bool(rng.integers(0, 4) == 0)
    use_skip = bool(rng.integers(0, 2))
    use_x2 = (not transposed) and (not three_d) and Ci % 8 == 0 and Ci >= 16 and bool(rng.integers(0, 2))
    shape = (Ci, D, H, W) if three_d else (Ci, H, W)
    x = torch.randn(shape, device="cuda")
    kd = (3, 3, 3) if three_d else (3, 3)
    if transposed:
        w = torch.randn((Ci, Co) + kd, device="cuda") * 0.2
    else:
        w = torch.randn((Co, Ci) + kd, device="cuda") * 0.2
    sc = torch.rand(Co, device="cuda") + 0.5
    sh = torch.randn(Co, device="cuda")
    outs = {}
    for path in ("mfma", "direct"):
        config.switches["D3D_CONV"] = path
        ops.clear_weight_cache()
        if transposed:
            f = ops.convtranspose3d_k3s2 if three_d else ops.convtranspose2d_k3s2
            probe = f(x, w, sc, sh, None, **({"relu": True} if three_d else {"act": 1}))
            skip = torch.randn_like(probe) if use_skip else None
            torch.manual_seed(case)
            skip = torch.randn(probe.shape, device="cuda", generator=None) if use_skip else None
            if use_skip:
                torch.manual_seed(case); skip = torch.randn(probe.shape, device="cuda")
            y = f(x, w, sc, sh, skip, **({"relu": True} if three_d else {"act": 1, "skip_after_act": True}))
        elif three_d:
            probe = ops.conv3d_k3(x, w, sc, sh, None, relu=True, stride=stride)
            if use_skip:
                torch.manual_seed(case); skip = torch.randn(probe.shape, device="cuda")
            else:
                skip = None
            y = ops.conv3d_k3(x, w, sc, sh, skip, relu=True, stride=stride)
        else:
            xa, xb = (x[: Ci // 2].contiguous(), x[Ci // 2:].contiguous()) if use_x2 else (x, None)
            probe = ops.conv2d_k3(xa, w, sc, sh, None, act=1, stride=stride, x2=xb)
            if use_skip:
                torch.manual_seed(case); skip = torch.randn(probe.shape, device="cuda")
            else:
                skip = None
            y = ops.conv2d_k3(xa, w, sc, sh, skip, act=1, stride=stride, x2=xb)
        outs[path] = y
    torch.cuda.synchronize()
    err = (outs["mfma"] - outs["direct"]).abs().max().item() / max(1.0, outs["direct"].abs().max().item())
    worst = max(worst, err)
    tag = "%s Ci=%d Co=%d %s s%d%s%s%s" % ("3D" if three_d else "2D", Ci, Co, tuple(shape[1:]), stride,
                                          " T" if transposed else "", " skip" if use_skip else "", " x2" if use_x2 else "")
    print("%-60s max-rel %.2e %s" % (tag, err, "" if err < 5e-5 else "  <-- MISMATCH"), flush=True)
print("worst %.2e over %d cases" % (worst, n_cases))
sys.exit(0 if worst < 5e-5 else 1)
