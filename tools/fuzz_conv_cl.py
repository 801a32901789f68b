"""Random-shape cross-check of the channel-last bf16 convolution kernels (stride 1 / stride 2 / transposed / C_out = 1, all the
(C_in, C_out) pairs of CostRegNet) against torch's fp32 convolutions on bf16-rounded operands (GPU box).
FUZZ_SEED, FUZZ_CASES as the other fuzzers."""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))
n_cases = int(os.environ.get("FUZZ_CASES", "60"))
torch.backends.cudnn.allow_tf32 = False
bf = lambda t: t.to(ops.h16_dtype()).float()
cl = lambda t: t.permute(1, 2, 3, 0).contiguous().to(ops.h16_dtype())      # planar fp32 [C,D,H,W] -> CL bf16
uncl = lambda t: t.float().permute(3, 0, 1, 2).contiguous()
S1 = [(8, 8), (16, 8), (32, 8), (16, 16), (32, 32), (64, 64), (8, 16), (32, 16), (8, 1), (16, 1), (32, 1)]
S2 = [(8, 16), (16, 32), (32, 64), (8, 8), (16, 16)]
T2 = [(16, 8), (32, 16), (64, 32), (16, 16)]
worst, nbad = 0.0, 0
ops.set_conv_precision("h16")
for case in range(n_cases):
    kind = str(rng.choice(["s1", "s2", "t2"]))
    Ci, Co = [S1, S2, T2][["s1", "s2", "t2"].index(kind)][int(rng.integers(0, [len(S1), len(S2), len(T2)][["s1", "s2", "t2"].index(kind)]))]
    D, H, W = int(rng.integers(1, 14)), int(rng.integers(1, 40)), int(rng.integers(1, 150))
    if Co == 1:
        W = 4 * max(1, W // 4)                 # planar fp32 output: W % 4 == 0
    x = torch.randn(Ci, D, H, W, device="cuda")
    relu = bool(rng.integers(0, 2)); use_skip = bool(rng.integers(0, 2)) and Co != 1
    sc = torch.rand(Co, device="cuda") + 0.5; sh = torch.randn(Co, device="cuda")
    if kind == "t2":
        wt = torch.randn(Ci, Co, 3, 3, 3, device="cuda") * 0.1
        ref = F.conv_transpose3d(bf(x)[None], bf(wt), stride=2, padding=1, output_padding=1)[0]
    else:
        wt = torch.randn(Co, Ci, 3, 3, 3, device="cuda") * 0.1
        ref = F.conv3d(bf(x)[None], bf(wt), stride=1 if kind == "s1" else 2, padding=1)[0]
    ref = ref * sc[:, None, None, None] + sh[:, None, None, None]
    if relu:
        ref = ref.clamp_min(0)
    sk = bf(torch.randn_like(ref)) if use_skip else None
    if use_skip:
        ref = ref + sk
    if kind == "t2":
        got = uncl(ops.convtranspose3d_k3s2_cl(cl(x), wt, sc, sh, cl(sk) if use_skip else None, relu=relu))
    elif Co == 1:
        in_cl = bool(rng.integers(0, 2))
        got = ops.conv3d_k3_cl(cl(x) if in_cl else x, wt, sc, sh, None, relu=relu, out_cl=False)
    else:
        in_cl = bool(rng.integers(0, 2)) or Ci == 64 or kind == "s2"
        got = uncl(ops.conv3d_k3_cl(cl(x) if in_cl else x, wt, sc, sh, cl(sk) if use_skip else None, relu=relu,
                                    stride=1 if kind == "s1" else 2))
    torch.cuda.synchronize()
    tol = 1e-4 * max(1.0, ref.abs().max().item())
    err = ((got - ref).abs() - (0.0 if Co == 1 else 2.0 ** -8) * ref.abs()).max().item()
    bad = (not torch.isfinite(got).all().item()) or err > tol or tuple(got.shape) != tuple(ref.shape)
    nbad += bad
    worst = max(worst, err / tol)
    print("%-3s %2d->%2d %2dx%2dx%3d relu=%d skip=%d  excess err / tol %.3f %s" % (kind, Ci, Co, D, H, W, relu, use_skip, err / tol,
                                                                                     "  <-- MISMATCH" if bad else ""), flush=True)
print("worst %.3f of the tolerance over %d cases, %d mismatches" % (worst, n_cases, nbad))
sys.exit(1 if nbad else 0)
