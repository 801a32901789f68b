"""Per-kernel time of one FeatureNet forward (one 2752x1856 image) -- the input side of every view."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import predict as P, synthetic as S

model = P.build_model(sys.argv[1] if len(sys.argv) > 1 else "casmvsnet", 384)
S.fill_state_dict_(model.state_dict(), 1)
net = model.cuda().eval().feature
x = torch.randn(1, 3, 1856, 2752, device="cuda")
with torch.no_grad():
    for _ in range(2):
        net(x)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        net(x)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=24, max_name_column_width=70))
