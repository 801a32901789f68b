"""Where does the stride-2 fused cell differ from the three launches?  (debugging aid)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
MODE = os.environ.get("GRU2_MODE", "")   # "" | "nocand" (wc = 0, bc = 0: h' = u h) | "nostate" (h = 0: h' = (1 - u) tanh(c)) | "nogate" (wg = 0: r = u = sigmoid(bg))
for (h, w) in [(272, 264), (135, 248), (64, 64)]:
    rng = np.random.default_rng(802)
    C, hid = 8, 16
    H, W = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    cost = dev(rng.standard_normal((C, h, w))); state = dev(rng.standard_normal((hid, H, W)))
    w1 = dev(rng.standard_normal((hid, C, 3, 3)) / np.sqrt(9 * C))
    wg = dev(rng.standard_normal((2 * hid, 2 * hid, 3, 3)) / np.sqrt(18 * hid)); wc = dev(rng.standard_normal((hid, 2 * hid, 3, 3)) / np.sqrt(18 * hid))
    bg, bc = dev(rng.standard_normal(2 * hid)), dev(rng.standard_normal(hid))
    if MODE == "nocand": wc.zero_(); bc.zero_()
    if MODE == "nostate": state.zero_()
    if MODE == "nogate": wg.zero_()
    with ops.h16_convs():
        outs = [ops.gru_cell_conv_fused(cost, state, w1, wg, bg, wc, bc, 2) for _ in range(3)]
        x = ops.conv2d_s2_zs(cost, w1, None, None, None, 1)
        gates = ops.conv2d_zs(x, wg, None, bg, state, 2, x2=state, ep_split=hid)
        want = ops.conv2d_zs(x, wc, None, bc, state, 3, x2=gates[:hid].contiguous(), aux1=gates[hid:].contiguous())
    torch.cuda.synchronize()
    for k, o in enumerate(outs):
        bad = (o != want)
        rows = bad.any(0).any(1).nonzero().flatten().tolist()
        cols = bad.any(0).any(0).nonzero().flatten().tolist()
        chans = bad.any(1).any(1).nonzero().flatten().tolist()
        print("%dx%d run %d: %d mismatches of %d; max diff %.3g; rows %s; cols %s..; channels %s" % (
            H, W, k, int(bad.sum()), bad.numel(), float((o - want).abs().max()), rows[:24], cols[:12], chans), flush=True)
