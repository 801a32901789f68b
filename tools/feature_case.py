"""A few FeatureNet forwards on a 2752x1856 image (workload for tools/run_pmc_script.sh; argv[1]: casmvsnet (default) | adamvs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import predict as P, synthetic as S
model = P.build_model(sys.argv[1] if len(sys.argv) > 1 else "casmvsnet", 384)
S.fill_state_dict_(model.state_dict(), 1)
net = model.cuda().eval().feature
x = torch.randn(1, 3, 1856, 2752, device="cuda")
with torch.no_grad():
    for _ in range(4):
        net(x)
torch.cuda.synchronize()
print("done")
