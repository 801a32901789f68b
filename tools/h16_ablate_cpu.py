#!/usr/bin/env python3
"""Which bf16 roundings of the CasMVSNet regulariser carry the depth error on the PEAKED 256 x 384 fixture?  

CPU experiment, build container only (imports the reference from /root/reference as tests/golden/make_golden.py does): the
reference's CostRegNet.forward is replaced by a restatement with a bf16 rounding at every site where bf16 mode rounds -- the
variance volume, each layer's weights, each layer's stored activation -- and each site can be switched back to fp32.  Prints the
mean depth error in stage-3 intervals per stage for a list of site sets.

    python tools/h16_ablate_cpu.py [casmvsnet]
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(os.environ.get("D3D_REFERENCE", "/root/reference"), "mvs", "mvs_cas"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from deep3d_aerial_amd import synthetic as S  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self
torch.set_num_threads(8)
from models import cas_mvsnet as RC  # noqa: E402

T = torch.from_numpy
LAYERS = ["conv0", "conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv9", "conv11", "prob"]
SITES = ["in"] + ["w_" + l for l in LAYERS] + ["a_" + l for l in LAYERS[:-1]]
ROUND = set()          # the sites rounded to bf16 in the current run
SPLIT2 = set()         # sites kept as a bf16 hi + lo pair (16 mantissa bits)
FMT = [torch.bfloat16]  # the 16-bit format of the rounded sites


def rnd(x, site):
    if site in SPLIT2:
        hi = x.to(torch.bfloat16).float()
        return hi + (x - hi).to(torch.bfloat16).float()
    if site not in ROUND:
        return x
    if FMT[0] == torch.float16:   # saturating, as the kernels convert
        x = x.clamp(-65504.0, 65504.0)
    return x.to(FMT[0]).float()


def block(m, x, name, skip=None):
    """ConvBnReLU3D / the transposed Sequential: conv (rounded weights) -> BN (fp32 epilogue) -> ReLU -> (+ skip) -> rounded store."""
    conv, bn = (m.conv, m.bn) if hasattr(m, "conv") else (m[0], m[1])
    w = rnd(conv.weight, "w_" + name)
    if isinstance(conv, torch.nn.ConvTranspose3d):
        y = F.conv_transpose3d(x, w, None, stride=conv.stride, padding=conv.padding, output_padding=conv.output_padding)
    else:
        y = F.conv3d(x, w, None, stride=conv.stride, padding=conv.padding)
    y = F.relu(F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps))
    if skip is not None:
        y = skip + y
    return rnd(y, "a_" + name)


def costreg_forward(self, x):
    x = rnd(x, "in")
    c0 = block(self.conv0, x, "conv0")
    c2 = block(self.conv2, block(self.conv1, c0, "conv1"), "conv2")
    c4 = block(self.conv4, block(self.conv3, c2, "conv3"), "conv4")
    y = block(self.conv6, block(self.conv5, c4, "conv5"), "conv6")
    y = block(self.conv7, y, "conv7", c4)
    y = block(self.conv9, y, "conv9", c2)
    y = block(self.conv11, y, "conv11", c0)
    return F.conv3d(y, rnd(self.prob.weight, "w_prob"), self.prob.bias, padding=1)


RC.CostRegNet.forward = costreg_forward


def main():
    H, W, V, nd, seed, gain = 256, 384, 5, 384, 7202, 20.0
    net = RC.Infer_CascadeMVSNet(num_depth=nd).eval()
    S.fill_state_dict_(net.state_dict(), seed)
    S.sharpen_state_dict_(net.state_dict(), gain)
    imgs, pm, dv = S.model_inputs(V, H, W, nd, seed)
    interval = float(dv[0, -1] - dv[0, 0]) / nd

    def run(rounded, split2=()):
        ROUND.clear(); ROUND.update(rounded)
        SPLIT2.clear(); SPLIT2.update(split2)
        with torch.no_grad():
            o = net(T(imgs), {k: T(v) for k, v in pm.items()}, T(dv))
        return [o[s]["depth"][0].numpy() for s in ("stage1", "stage2", "stage3")]

    ref = run(())
    g = np.load(os.path.join(ROOT, "tests", "golden", "model_casmvsnet_v5_256_peaked.npz"))
    print("restatement vs golden (fp32): %.2e intervals" % (np.abs(ref[2] - g["stage3_depth"]).mean() / interval))

    def report(label, rounded, split2=()):
        d = run(rounded, split2)
        e = [np.abs(a - b).mean() / interval for a, b in zip(d, ref)]
        rl1 = np.abs(d[2] - ref[2]).mean() / np.abs(ref[2]).mean()
        print("%-58s stage errors %.3f %.3f %.3f intervals, final rel-L1 %.2e" % (label, e[0], e[1], e[2], rl1), flush=True)

    allr = set(SITES)
    if "--f16" in sys.argv:
        FMT[0] = torch.float16
        report("all sites IEEE half (f16), fp32 accumulate", allr)
        report("f16: only the input volume rounded", {"in"})
        report("f16: only weights rounded", {s for s in SITES if s.startswith("w_")})
        report("f16: only activations rounded", {s for s in SITES if s.startswith("a_")})
        return
    report("all bf16 (the mode as shipped)", allr)
    report("only the input volume rounded", {"in"})
    report("only weights rounded", {s for s in SITES if s.startswith("w_")})
    report("only activations rounded", {s for s in SITES if s.startswith("a_")})
    for s in SITES:
        report("all but %s" % s, allr - {s})
    for s in SITES:
        report("only %s" % s, {s})
    full = {"in", "w_conv0", "a_conv0", "w_conv11", "a_conv11", "w_prob"}
    report("full-resolution sites fp32, the rest bf16", allr - full)
    report("full-resolution sites bf16, the rest fp32", full)
    report("prob + conv11 output fp32", allr - {"w_prob", "a_conv11"})
    report("prob, conv11 out, conv0 out fp32", allr - {"w_prob", "a_conv11", "a_conv0"})
    report("prob, conv11 out, conv0 out, in fp32", allr - {"w_prob", "a_conv11", "a_conv0", "in"})
    report("all sites as hi + lo pairs", (), allr)


if __name__ == "__main__":
    main()
