#!/usr/bin/env python3
"""The conv-GRU cells of AdaMVS's slice regulariser at the three cascade shapes (bf16 mode): one fused launch
(csrc/gru_fused.hip) against the three tile-kernel launches it replaces, with the HBM bytes a cell has to move.

    python tools/gru_fused_bench.py [reps]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import config, ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def timed(fn):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3   # us


# (label, C, stride, cost h, cost w)
cases = [("stage 1 gru1", 32, 1, 688, 464), ("stage 1 gru2", 8, 2, 688, 464), ("stage 2 gru1", 16, 1, 1376, 928),
         ("stage 2 gru2", 8, 2, 1376, 928), ("stage 3 gru1", 8, 1, 2752, 1856), ("stage 3 gru2", 8, 2, 2752, 1856)]
print("%-14s %10s %10s %8s %10s %8s" % ("cell", "fused us", "3 launches", "ratio", "min MB", "fused TB/s"))
with ops.h16_convs():
    for label, C, s, h, w in cases:
        hid = 8 if s == 1 else 16
        H, W = (h, w) if s == 1 else ((h - 1) // 2 + 1, (w - 1) // 2 + 1)
        cost, st = dev(rng.standard_normal((C, h, w))), dev(rng.standard_normal((hid, H, W)))
        w1 = dev(rng.standard_normal((hid, C, 3, 3)) / np.sqrt(9 * C))
        wg = dev(rng.standard_normal((2 * hid, 2 * hid, 3, 3)) / np.sqrt(18 * hid))
        wc = dev(rng.standard_normal((hid, 2 * hid, 3, 3)) / np.sqrt(18 * hid))
        bg, bc = dev(rng.standard_normal(2 * hid)), dev(rng.standard_normal(hid))

        def fused():
            return ops.gru_cell_conv_fused(cost, st, w1, wg, bg, wc, bc, s)

        def separate():
            x = ops.conv2d_zs(cost, w1, None, None, None, 1) if s == 1 else ops.conv2d_s2_zs(cost, w1, None, None, None, 1)
            g = ops.conv2d_zs(x, wg, None, bg, st, 2, x2=st, ep_split=hid)
            return ops.conv2d_zs(x, wc, None, bc, st, 3, x2=g[:hid], aux1=g[hid:])

        tf, ts = timed(fused), timed(separate)
        mb = (cost.numel() + 2 * st.numel()) * 4 / 1e6
        print("%-14s %10.1f %10.1f %8.2f %10.1f %8.2f" % (label, tf, ts, ts / tf, mb, mb / tf))
