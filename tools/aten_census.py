#!/usr/bin/env python3
"""ATen operators (and memcpys) issued by one Infer_* forward, by count: what is left of torch glue between the C-ABI launches.
    D3D_CONV_PRECISION=bf16 python tools/aten_census.py adamvs [H W]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import predict, synthetic as S

model = sys.argv[1] if len(sys.argv) > 1 else "adamvs"
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2752, 1856)
net = predict.build_model(model, 384); S.fill_state_dict_(net.state_dict(), 1); net = net.cuda().eval()
s = predict.SyntheticBlock(1, 5, H, W, 384)[0]
imgs = torch.from_numpy(s["imgs"])[None].cuda()
pm = {k: torch.from_numpy(v)[None].cuda() for k, v in s["proj_matrices"].items()}
dv = torch.from_numpy(s["depth_values"])[None].cuda()
from torch.profiler import profile, ProfilerActivity
with torch.no_grad():
    net(imgs, pm, dv); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        net(imgs, pm, dv); torch.cuda.synchronize()
from torch.autograd import DeviceType
dev_ev = [e for e in prof.events() if e.device_type == DeviceType.CUDA]
kern = [e for e in dev_ev if not e.name.startswith("Memcpy") and not e.name.startswith("Memset")]
print("device events in one warm forward: %d kernels, %d memcpy / memset; GPU time %.2f ms" % (
    len(kern), len(dev_ev) - len(kern), sum(e.device_time for e in dev_ev) / 1e3))
ev = [e for e in prof.key_averages(group_by_stack_n=4) if e.key.startswith("aten::") or "Memcpy" in e.key or "copyBuffer" in e.key]
ev.sort(key=lambda e: -e.count)
for e in ev[:25]:
    st = [f for f in (e.stack or []) if "deep3d_aerial_amd" in f]
    print("%5d x %-28s %s" % (e.count, e.key[:28], (st[0].split("deep3d_aerial_amd/")[-1] if st else "")[:90]))
