#!/usr/bin/env python3
"""CPU model of the sweep kernel's ring planning (no GPU): LDS floats the rings of a workgroup need.

Mirrors candidate_windows / plan_tables of csrc/planesweep_tiled.hip: per (patch, depth segment, step) the window of
a view is the hull of the 8 projected corners; RW x RH is the largest union of consecutive windows; a view's ring
takes pitch(RW) * (RH + 1) words.  Prints, per scene and patch shape, the share of workgroups whose rings fit the
capacity at each step size -- the planes-per-step the kernel would pick and its fallback share.

    python tools/ring_sim.py [config2|config5] [samples]
"""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import synthetic as S  # noqa: E402


def pitch(RW, stride):
    n = (RW + 1) * stride
    return ((n - 32 + 63) & ~63) + 32


def windows(M, x0, y0, x1, y1, dlo, dhi, h, w):
    us, vs = [], []
    for cx in (x0, x1):
        for cy in (y0, y1):
            r = M[:3, 0] * cx + M[:3, 1] * cy + M[:3, 2]
            for d in (dlo, dhi):
                p = r * d + M[:3, 3]
                us.append(np.clip(p[0] / p[2], -8, w + 8))
                vs.append(np.clip(p[1] / p[2], -8, h + 8))
    wx0 = max(int(np.floor(min(us) - 0.0625)), -1)
    wy0 = max(int(np.floor(min(vs) - 0.0625)), -1)
    wx1 = min(int(np.floor(max(us) + 0.0625)) + 1, w)
    wy1 = min(int(np.floor(max(vs) + 0.0625)) + 1, h)
    ww, wh = max(wx1 - wx0 + 1, 0), max(wy1 - wy0 + 1, 0)
    if ww < 2 or wh < 2:
        return None
    return wx0, wy0, ww, wh


def need_words(Ms, x0, y0, tw, th, depths, sp, h, w, stride, quant=True):
    x1, y1 = min(x0 + tw - 1, w - 1), min(y0 + th - 1, h - 1)
    total = 0
    for M in Ms:
        prev = None
        RW = RH = 1
        for k in range(0, len(depths), sp):
            dd = depths[k:k + sp]
            win = windows(M, x0, y0, x1, y1, dd.min(), dd.max(), h, w)
            if win is None:
                prev = None
                continue
            ux, uy = win[2], win[3]
            if prev is not None:
                ux = max(win[0] + win[2], prev[0] + prev[2]) - min(win[0], prev[0])
                uy = max(win[1] + win[3], prev[1] + prev[3]) - min(win[1], prev[1])
            RW, RH = max(RW, ux), max(RH, uy)
            prev = win
        if quant:
            RW = (RW + 7) // 8 * 8
            RH = (RH + 1) // 2 * 2
        total += pitch(RW, stride) * (RH + 1)
    return total


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "config2"
    nsamp = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    if cfg == "config2":
        V, h, w, D, seed = 5, 688, 464, 384, 0
    else:
        V, h, w, D, seed = 7, 928, 688, 512, 5
    proj, dv = S.make_scene(V, h, w, D, seed=seed)
    depths = S.uniform_depths(dv, D).astype(np.float64)
    P = proj.astype(np.float64)
    Ms = [P[i] @ np.linalg.inv(P[0]) for i in range(1, V)]
    rng = np.random.default_rng(0)
    dseg = 128
    print("%s: %d source views, features %dx%d, D=%d, depth segments of %d planes" % (cfg, V - 1, h, w, D, dseg))
    # (label, patch w, patch h, position stride in 4-byte words, table words)
    nsrc = V - 1
    tables = 2 * 36 + 2 * 128 + 32 + 128 + 64 * nsrc * 8 + 140 + 257 * 8
    shapes = [("32x4 fp32 x16ch (20 w)", 32, 4, 20), ("32x4 fp32 x8ch / fp16 x16ch (12 w)", 32, 4, 12),
              ("32x4 fp16 x8ch (4 w)", 32, 4, 4), ("32x2 fp16 x16ch (12 w)", 32, 2, 12), ("16x4 fp16 x16ch (12 w)", 16, 4, 12),
              ("32x4 fp32 x32ch (36 w)", 32, 4, 36), ("16x4 fp32 x32ch (36 w)", 16, 4, 36), ("32x2 fp32 x32ch (36 w)", 32, 2, 36)]
    cap = 40960 - tables
    for label, tw, th, stride in shapes:
        fits = {sp: 0 for sp in (16, 12, 8, 4, 2)}
        best = []
        for _ in range(nsamp):
            x0 = int(rng.integers(0, (w + tw - 1) // tw)) * tw
            y0 = int(rng.integers(0, (h + th - 1) // th)) * th
            seg = int(rng.integers(0, D // dseg))
            dd = depths[seg * dseg:(seg + 1) * dseg]
            got = 0
            for sp in (16, 12, 8, 4, 2):
                n = need_words(Ms, x0, y0, tw, th, dd, sp, h, w, stride)
                if n > cap:
                    n = need_words(Ms, x0, y0, tw, th, dd, sp, h, w, stride, quant=False)
                if n <= cap:
                    fits[sp] += 1
                    got = max(got, sp)
            best.append(got)
        best = np.array(best)
        print("  %-40s cap %5d words | fits at SP=16/12/8/4/2: %s | mean best SP %.1f | fallback %.1f %%" % (
            label, cap, " ".join("%3.0f%%" % (100.0 * fits[sp] / nsamp) for sp in (16, 12, 8, 4, 2)), best.mean(), 100.0 * (best == 0).mean()))


if __name__ == "__main__":
    main()
