"""A few regulariser layers at full cascade size, twice each: the workload profiled by tools/profile_conv.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops  # noqa: E402

H, W = 1856, 2752
x = torch.randn(8, 8, H, W, device="cuda")
w = torch.randn(8, 8, 3, 3, 3, device="cuda") * 0.1
w1 = torch.randn(1, 8, 3, 3, 3, device="cuda") * 0.1
x2 = torch.randn(16, 4, H // 2, W // 2, device="cuda")
w2 = torch.randn(16, 16, 3, 3, 3, device="cuda") * 0.1
wt = torch.randn(16, 8, 3, 3, 3, device="cuda") * 0.1
for _ in range(2):
    ops.conv3d_k3(x, w)                 # stage-3 conv0  8->8   conv_stream_kernel<1,4,3,8>
    ops.conv3d_k3(x, w1, relu=False)    # stage-3 prob   8->1   conv_stream_kernel<1,1,3,8>
    ops.conv3d_k3(x2, w2)               # stage-3 conv2 16->16  conv_stream_kernel<1,4,3,16>
    ops.convtranspose3d_k3s2(x2, wt)    # stage-3 conv11 16->8  conv_stream_kernel<4,4,2,16>
torch.cuda.synchronize()
