"""The 2-D slice-regulariser convolutions of AdaMVS at the stage-3 and stage-2 slice shapes (GPU box; workload for
tools/run_pmc_script.sh).  argv[1]: bf16 (default) | fp32 -- the precision mode the tile kernels are picked for."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops
ops.set_conv_precision(sys.argv[1] if len(sys.argv) > 1 else "h16")
H, W = 1856, 2752
x = torch.randn(8, H, W, device="cuda"); h8 = torch.randn(8, H, W, device="cuda")
wg = torch.randn(16, 16, 3, 3, device="cuda") * 0.1; bg = torch.randn(16, device="cuda")
wc = torch.randn(8, 16, 3, 3, device="cuda") * 0.1; bc = torch.randn(8, device="cuda")
for _ in range(4):
    ops.gru_cell_fused(x, h8, wg, bg, wc, bc)
x2 = torch.randn(16, H // 2, W // 2, device="cuda"); h16 = torch.randn(16, H // 2, W // 2, device="cuda")
wg2 = torch.randn(32, 32, 3, 3, device="cuda") * 0.1; bg2 = torch.randn(32, device="cuda")
wc2 = torch.randn(16, 32, 3, 3, device="cuda") * 0.1; bc2 = torch.randn(16, device="cuda")
for _ in range(4):
    ops.gru_cell_fused(x2, h16, wg2, bg2, wc2, bc2)
torch.cuda.synchronize()
print("done")
