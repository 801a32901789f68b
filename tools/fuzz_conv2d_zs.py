"""Random-shape cross-check of the 2-D tile kernels of the slice regularisers (stride 1 with the two-input concat and the ReLU /
GRU gate / GRU update epilogues, stride 2, transposed) against torch's fp32 convolutions (on bf16-rounded operands in bf16 mode) (GPU box).
FUZZ_SEED, FUZZ_CASES as the other fuzzers; FUZZ_PRECISION=bf16|fp32 selects the kernels' precision."""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
n_cases = int(os.environ.get("FUZZ_CASES", "80"))
ops.set_conv_precision(os.environ.get("FUZZ_PRECISION", "h16"))
torch.backends.cudnn.allow_tf32 = False
bf = (lambda t: t.to(ops.h16_dtype()).float()) if os.environ.get("FUZZ_PRECISION", "h16") == "h16" else (lambda t: t)   # fp32 mode: exact operands
nbad, worst = 0, 0.0
for case in range(n_cases):
    kind = str(rng.choice(["s1", "gates", "update", "s2", "t2"]))
    H = int(rng.integers(1, 60))
    W = 4 * int(rng.integers(1, 80))
    relu = int(rng.integers(0, 2))
    if kind in ("s1", "gates", "update"):
        C0, C1 = [(8, 0), (16, 0), (32, 0), (8, 8), (16, 16), (8, 24), (24, 8)][int(rng.integers(0, 7))]
        if kind != "s1" and C1 == 0:
            C0, C1 = 8, 8
        Hc = C1 if kind != "s1" else 0
        Co = {"s1": int(rng.choice([1, 8, 16, 32])), "gates": 2 * Hc, "update": Hc}[kind]
        if Co > 32:
            C0, C1, Hc = 8, 8, 8
            Co = 16 if kind == "gates" else 8
        x = torch.randn(C0, H, W, device="cuda"); x2 = torch.randn(C1, H, W, device="cuda") if C1 else None
        w = torch.randn(Co, C0 + C1, 3, 3, device="cuda") * 0.1; b = torch.randn(Co, device="cuda")
        xin = x if x2 is None else torch.cat([x, x2], 0)
        conv = F.conv2d(bf(xin)[None], bf(w), padding=1)[0] + b[:, None, None]
        if kind == "s1":
            sk = torch.randn(Co, H, W, device="cuda"); after = bool(rng.integers(0, 2))
            y = conv + (0 if after else sk)
            y = y.clamp_min(0) if relu else y
            ref = y + (sk if after else 0)
            got = ops.conv2d_zs(x, w, None, b, sk, relu, x2=x2, skip_after_act=after)
        elif kind == "gates":
            h = x2
            sg = torch.sigmoid(conv)
            ref = torch.cat([sg[:Hc] * h, sg[Hc:]], 0)
            got = ops.conv2d_zs(x, w, None, b, h, 2, x2=x2, ep_split=Hc)
        else:
            h = torch.randn(Co, H, W, device="cuda"); u = torch.rand(Co, H, W, device="cuda")
            ref = u * h + (1 - u) * torch.tanh(conv)
            got = ops.conv2d_zs(x, w, None, b, h, 3, x2=x2, aux1=u)
        tag = "%-6s %2d+%2d->%2d" % (kind, C0, C1, Co)
    elif kind == "s2":
        Ci, Co = [(8, 16), (16, 32), (8, 8), (16, 16), (8, 1)][int(rng.integers(0, 5))]
        W = 8 * int(rng.integers(1, 40))
        x = torch.randn(Ci, H, W, device="cuda"); w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.1; b = torch.randn(Co, device="cuda")
        conv = F.conv2d(bf(x)[None], bf(w), stride=2, padding=1)[0] + b[:, None, None]
        ref = conv.clamp_min(0) if relu else conv
        got = ops.conv2d_s2_zs(x, w, None, b, None, relu)
        tag = "%-6s %2d->%2d   " % (kind, Ci, Co)
    else:
        Ci, Co = [(16, 8), (8, 1), (32, 16), (16, 16), (8, 8)][int(rng.integers(0, 5))]
        x = torch.randn(Ci, H, W, device="cuda"); w = torch.randn(Ci, Co, 3, 3, device="cuda") * 0.1; b = torch.randn(Co, device="cuda")
        conv = F.conv_transpose2d(bf(x)[None], bf(w), stride=2, padding=1, output_padding=1)[0] + b[:, None, None]
        sk = torch.randn_like(conv)
        y = conv + sk
        ref = y.clamp_min(0) if relu else y
        got = ops.convtranspose2d_zs(x, w, None, b, sk, act=relu, skip_after_act=False)
        tag = "%-6s %2d->%2d   " % (kind, Ci, Co)
    torch.cuda.synchronize()
    if got is None:
        print("%s %3dx%3d  (not taken)" % (tag, H, W), flush=True)
        continue
    tol = 1e-4 * max(1.0, conv.abs().max().item())
    err = (got - ref).abs().max().item()
    bad = (not torch.isfinite(got).all().item()) or err > tol or tuple(got.shape) != tuple(ref.shape)
    nbad += bad
    worst = max(worst, err / tol)
    print("%s %3dx%3d relu=%d  err / tol %.3f %s" % (tag, H, W, relu, err / tol, "  <-- MISMATCH" if bad else ""), flush=True)
print("worst %.3f of the tolerance over %d cases, %d mismatches" % (worst, n_cases, nbad))
sys.exit(1 if nbad else 0)
