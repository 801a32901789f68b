#!/usr/bin/env python3
"""Random-shape cross-checks of round 4's kernels (bf16 mode): the fused conv-GRU cell against the three tile-kernel launches
(bit for bit) and against float64 on rounded operands where the tile kernels do not take the shape; the wide convolution and
the 24- / 40-channel tile kernel against float64 on rounded operands; the fused head + regression against the two launches; conv11 + prob of a CostRegNet in one kernel against its two launches
(bit for bit); fp32 mode: the stride-2 split-operand 3-D layer against float64.
    python tools/fuzz_round4.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
bf = lambda t: t.to(ops.h16_dtype()).double()
bad = 0


def cell64(cost, state, w1, wg, bg, wc, bc, stride, hid):
    x = bf(F.relu(F.conv2d(bf(cost)[None], bf(w1), stride=stride, padding=1)).float())
    g = torch.sigmoid(F.conv2d(torch.cat([x, bf(state)[None]], 1), bf(wg), bg.double(), padding=1))
    rh = bf((g[:, :hid] * state.double()[None]).float())
    c = torch.tanh(F.conv2d(torch.cat([x, rh], 1), bf(wc), bc.double(), padding=1))
    u = g[:, hid:]
    return (u * state.double()[None] + (1 - u) * c)[0]


with ops.h16_convs():
    for i in range(cases):
        # ---- fused conv-GRU cell ----
        stride = int(rng.integers(1, 3))
        C = int(rng.choice([8, 16, 32])) if stride == 1 else 8
        hid = 8 if stride == 1 else 16
        W = 4 * int(rng.integers(2, 70))
        H = int(rng.integers(3, 150))
        h, w = (H, W) if stride == 1 else (2 * H - int(rng.integers(0, 2)), 2 * W)
        cost, st = dev(rng.standard_normal((C, h, w))), dev(rng.standard_normal((hid, H, W)))
        w1 = dev(rng.standard_normal((hid, C, 3, 3)) / np.sqrt(9 * C))
        wg = dev(rng.standard_normal((2 * hid, 2 * hid, 3, 3)) / np.sqrt(18 * hid))
        wc = dev(rng.standard_normal((hid, 2 * hid, 3, 3)) / np.sqrt(18 * hid))
        bg, bc = dev(rng.standard_normal(2 * hid)), dev(rng.standard_normal(hid))
        got = ops.gru_cell_conv_fused(cost, st, w1, wg, bg, wc, bc, stride)
        if got is None:
            print("cell case %d not taken: C %d stride %d %dx%d" % (i, C, stride, h, w)); bad += 1
        else:
            x = ops.conv2d_zs(cost, w1, None, None, None, 1) if stride == 1 else ops.conv2d_s2_zs(cost, w1, None, None, None, 1)
            if x is not None:
                g = ops.conv2d_zs(x, wg, None, bg, st, 2, x2=st, ep_split=hid)
                want = ops.conv2d_zs(x, wc, None, bc, st, 3, x2=g[:hid].contiguous(), aux1=g[hid:].contiguous())
                if not torch.equal(got, want):
                    print("cell case %d DIFFERS from the three launches: C %d stride %d %dx%d max %g" % (i, C, stride, h, w, float((got - want).abs().max()))); bad += 1
            ref = cell64(cost, st, w1, wg, bg, wc, bc, stride, hid)
            d = (got.double() - ref).abs()
            if float(d.mean()) > 3e-5 or float(d.max()) > 5e-3 or not torch.isfinite(got).all():
                print("cell case %d vs float64: C %d stride %d %dx%d mean %g max %g" % (i, C, stride, h, w, float(d.mean()), float(d.max()))); bad += 1
        # ---- wide convolution / 24-40 channel tile kernel ----
        C1, C2, Co = [(32, 32, 64), (32, 32, 32), (64, 64, 128), (64, 64, 64), (64, 0, 32), (16, 8, 16), (32, 8, 16), (32, 8, 8)][int(rng.integers(0, 8))]
        Hh, Ww = int(rng.integers(2, 100)), (int(rng.integers(2, 100)) if C1 + C2 >= 64 else 4 * int(rng.integers(2, 40)))
        x = dev(rng.standard_normal((C1, Hh, Ww)))
        x2 = dev(rng.standard_normal((C2, Hh, Ww))) if C2 else None
        wt = dev(rng.standard_normal((Co, C1 + C2, 3, 3)) / np.sqrt(9 * (C1 + C2)))
        bias, act = dev(rng.standard_normal(Co)), int(rng.integers(0, 2))
        got = ops.conv2d_wide(x, wt, None, bias, None, act, x2=x2) if C1 + C2 >= 64 else ops.conv2d_zs(x, wt, None, bias, None, act, x2=x2)
        xin = bf(x) if x2 is None else torch.cat([bf(x), bf(x2)])
        want = F.conv2d(xin[None], bf(wt), bias.double(), padding=1)[0]
        want = F.relu(want) if act else want
        if got is None or float((got.double() - want).abs().max()) > 3e-5 * max(1.0, float(want.abs().max())):
            print("conv case %d: %d+%d -> %d %dx%d act %d: %s" % (i, C1, C2, Co, Hh, Ww, act, "not taken" if got is None else "max %g" % float((got.double() - want).abs().max()))); bad += 1
        # ---- fused head + regression ----
        tr = bool(rng.integers(0, 2))
        hh, ww = int(rng.integers(2, 90)), (2 if tr else 4) * int(rng.integers(1, 50))
        up = dev(rng.standard_normal((8, hh, ww)))
        wh = dev(0.3 * rng.standard_normal((8, 1, 3, 3) if tr else (1, 8, 3, 3)))
        bh = dev(rng.standard_normal(1))
        HH, WW = (2 * hh, 2 * ww) if tr else (hh, ww)
        mode = int(rng.integers(0, 3))
        dpl = dev(600 + 50 * rng.standard_normal((1, 1) if mode == 0 else (hh, ww) if mode == 1 else (HH, WW)))
        acc0 = [dev(np.abs(rng.standard_normal((HH, WW)))) for _ in range(3)]
        a = [t.clone() for t in acc0]
        ok = ops.slice_head_regress(up, wh, bh, tr, dpl, *a)
        # the head evaluated in float64 on bf16-rounded operands, then the regression update through the separate kernel (the
        # two-launch path itself rounds its operands to bf16 only on the tile kernels, i.e. for widths that are multiples of 4)
        reg64 = (F.conv_transpose2d(bf(up)[None], bf(wh), bh.double(), stride=2, padding=1, output_padding=1) if tr
                 else F.conv2d(bf(up)[None], bf(wh), bh.double(), padding=1))[0]
        b = [t.clone() for t in acc0]
        ops.online_regress_update(reg64[0].float().contiguous(), dpl, *b)
        if not ok or any(float((p - q).abs().max()) > 3e-5 * float(q.abs().max()) for p, q in zip(a, b)):
            print("head case %d: transposed %s %dx%d dplane mode %d: %s" % (i, tr, hh, ww, mode, "not taken" if not ok else
                  "differs %s" % [float((p - q).abs().max() / q.abs().max()) for p, q in zip(a, b)])); bad += 1
        if ww % 4 == 0:
            b = [t.clone() for t in acc0]
            reg = ops.convtranspose2d_k3s2(up, wh, None, bh, None, act=0) if tr else ops.conv2d_k3(up, wh, None, bh, None, act=0)
            ops.online_regress_update(reg[0], dpl, *b)
            if ok and any(float((p - q).abs().max()) > 2e-5 * float(q.abs().max()) for p, q in zip(a, b)):
                print("head case %d vs the two launches: transposed %s %dx%d mode %d differs" % (i, tr, hh, ww, mode)); bad += 1
        # ---- conv11 + prob of a CostRegNet in one kernel against the two launches (bit for bit) ----
        D3, H3, W3 = int(rng.integers(1, 10)), int(rng.integers(1, 40)), 2 * int(rng.integers(1, 50))
        x3 = dev(rng.standard_normal((D3, H3, W3, 16))).to(ops.h16_dtype())
        sk3 = dev(rng.standard_normal((2 * D3, 2 * H3, 2 * W3, 8))).to(ops.h16_dtype()) if rng.integers(0, 4) else None
        w11, wp3 = dev(0.1 * rng.standard_normal((16, 8, 3, 3, 3))), dev(0.1 * rng.standard_normal((1, 8, 3, 3, 3)))
        sc3, sh3, bp3 = dev(rng.uniform(0.5, 1.5, 8)), dev(rng.standard_normal(8)), dev(rng.standard_normal(1))
        one = ops.convtranspose3d_prob_cl(x3, w11, sc3, sh3, sk3, wp3, bp3)
        y3 = ops.convtranspose3d_k3s2_cl(x3, w11, sc3, sh3, sk3, relu=True)
        two = ops.conv3d_k3_cl(y3, wp3, None, bp3, None, relu=False, stride=1, out_cl=False)[0]
        if one is None or not torch.equal(one, two):
            print("conv11 + prob case %d: %dx%dx%d skip %s: %s" % (i, D3, H3, W3, sk3 is not None, "not taken" if one is None else
                  "differs by %g" % float((one - two).abs().max()))); bad += 1
        # ---- upconv1 + skip + head + regression update in one kernel against the two launches (bit for bit) ----
        h2, w2 = int(rng.integers(1, 60)), 4 * int(rng.integers(1, 30))
        s2, s1 = dev(rng.standard_normal((16, h2, w2))), dev(rng.standard_normal((8, 2 * h2, 2 * w2)))
        wu, bu = dev(0.2 * rng.standard_normal((16, 8, 3, 3))), dev(rng.standard_normal(8))
        wh2, bh2 = dev(0.3 * rng.standard_normal((8, 1, 3, 3))), dev(rng.standard_normal(1))
        mode = int(rng.integers(0, 3))
        table = dev(600 + 50 * rng.standard_normal(8))
        dpl = table[int(rng.integers(0, 8))].view(1, 1) if mode == 0 else \
            dev(600 + 50 * rng.standard_normal((2 * h2, 2 * w2) if mode == 1 else (4 * h2, 4 * w2)))
        acc0 = [dev(np.abs(rng.standard_normal((4 * h2, 4 * w2)))) for _ in range(3)]
        a, b = [t.clone() for t in acc0], [t.clone() for t in acc0]
        ok = ops.slice_tail_regress(s2, wu, bu, s1, wh2, bh2, dpl, *a)
        upm = ops.convtranspose2d_k3s2(s2, wu, None, bu, s1, skip_after_act=False, act=1)
        ok2 = ops.slice_head_regress(upm, wh2, bh2, True, dpl, *b)
        if not ok or not ok2 or any(not torch.equal(p, q) for p, q in zip(a, b)):
            print("slice tail case %d: %dx%d dplane mode %d: %s" % (i, h2, w2, mode, "not taken" if not (ok and ok2) else "differs")); bad += 1
print("%d cases, %d problems" % (cases, bad))
# ---- fp32 mode: the stride-2 split-operand layer against float64 ----
for i in range(max(cases // 3, 1)):
    Ci, Co = [(8, 16), (16, 32), (32, 64)][int(rng.integers(0, 3))]
    D3, H3, W3 = int(rng.integers(1, 9)), int(rng.integers(1, 40)), int(rng.integers(1, 20)) * 8 - int(rng.integers(0, 2))
    x = dev(rng.standard_normal((Ci, D3, H3, W3)))
    wt = dev(rng.standard_normal((Co, Ci, 3, 3, 3)) / np.sqrt(27 * Ci))
    got = ops.conv3d_k3(x, wt, relu=False, stride=2)
    want = F.conv3d(x.double()[None], wt.double(), stride=2, padding=1)[0]
    err = float((got.double() - want).abs().max())
    if err > 4e-6 * max(1.0, float(want.abs().max())):
        print("stride-2 split case %d: %d -> %d %dx%dx%d max %g" % (i, Ci, Co, D3, H3, W3, err)); bad += 1
print("fp32 stride-2 cases %d, problems so far %d" % (max(cases // 3, 1), bad))
# ---- conv0 of a feature trunk in one launch against the two launches (bit for bit; fp32 precision of the feature nets) ----
with ops.fp32_convs():
    for i in range(max(cases // 6, 1)):
        H0, W0 = int(rng.integers(256, 420)), 4 * int(rng.integers(64, 130))
        img = dev(rng.uniform(0, 1, (3, H0, W0)))
        wa, wb = dev(0.4 * rng.standard_normal((8, 3, 3, 3))), dev(0.2 * rng.standard_normal((8, 8, 3, 3)))
        sa, ta, sb, tb = [dev(rng.standard_normal(8) * 0.3 + o) for o in (1.0, 0.0, 1.0, 0.0)]
        a0, a1 = int(rng.integers(0, 2)), int(rng.integers(0, 2))
        one = ops.conv2d_k3_pair3(img, wa, sa, ta, a0, wb, sb, tb, a1)
        two = ops.conv2d_k3(ops.conv2d_k3(img, wa, sa, ta, None, act=a0), wb, sb, tb, None, act=a1)
        if one is None or not torch.equal(one, two):
            print("conv0 pair case %d: %dx%d act %d %d: %s" % (i, H0, W0, a0, a1, "not taken" if one is None else "differs by %g" % float((one - two).abs().max()))); bad += 1
print("conv0 pair cases %d, problems so far %d" % (max(cases // 6, 1), bad))
sys.exit(1 if bad else 0)
