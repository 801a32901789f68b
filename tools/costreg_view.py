"""The three CostRegNets of a CasMVSNet view, bf16 mode, on channel-last volumes as the sweep kernel leaves them (GPU box;
what tools/run_pmc_hbm.sh profiles for profiles/r02_costreg_hbm_traffic.txt)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops, synthetic as S
from deep3d_aerial_amd.cas_mvsnet import CostRegNet
ops.set_conv_precision("h16")
for C, D, h, w in ((32, 48, 464, 688), (16, 32, 928, 1376), (8, 8, 1856, 2752)):
    net = CostRegNet(C).cuda().eval()
    S.fill_state_dict_(net.state_dict(), 3)
    vol = torch.randn(D, h, w, C, device="cuda").to(ops.h16_dtype())
    with torch.no_grad():
        for _ in range(2):
            net.forward_one(vol)
    torch.cuda.synchronize()
    del net, vol
    torch.cuda.empty_cache()
print("done")
