"""One fused conv-GRU cell at the stage-3 shape, a few launches (for tools/pmc_kernel.sh)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops
rng = np.random.default_rng(0)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
C, hid, h, w = 8, 8, 2752, 1856
cost, st = dev(rng.standard_normal((C, h, w))), dev(rng.standard_normal((hid, h, w)))
w1 = dev(rng.standard_normal((hid, C, 3, 3)) / np.sqrt(9 * C))
wg = dev(rng.standard_normal((2 * hid, 2 * hid, 3, 3)) / np.sqrt(18 * hid))
wc = dev(rng.standard_normal((hid, 2 * hid, 3, 3)) / np.sqrt(18 * hid))
bg, bc = dev(rng.standard_normal(2 * hid)), dev(rng.standard_normal(hid))
with ops.h16_convs():
    for _ in range(3):
        ops.gru_cell_conv_fused(cost, st, w1, wg, bg, wc, bc, 1)
torch.cuda.synchronize()
