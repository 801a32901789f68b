#!/usr/bin/env python3
"""Times full Infer_* forwards at BASELINE image size (2752x1856, V=5) with seeded random weights and
prints the per-kernel breakdown (torch profiler). Not part of bench.py's headline metric."""
import argparse, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import predict, synthetic as S

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="casmvsnet")
ap.add_argument("--h", type=int, default=2752)
ap.add_argument("--w", type=int, default=1856)
ap.add_argument("--views", type=int, default=5)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
net = predict.build_model(a.model, 384)
S.fill_state_dict_(net.state_dict(), 1)
net = net.cuda().eval()
ds = predict.SyntheticBlock(1, a.views, a.h, a.w, 384)
s = ds[0]
imgs = torch.from_numpy(s["imgs"])[None].cuda()
pm = {k: torch.from_numpy(v)[None].cuda() for k, v in s["proj_matrices"].items()}
dv = torch.from_numpy(s["depth_values"])[None].cuda()
with torch.no_grad():
    net(imgs, pm, dv); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps): out = net(imgs, pm, dv)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    print("%s %dx%d V=%d: %.1f ms per reference view; peak mem %.1f GB" % (a.model, a.h, a.w, a.views, dt * 1e3, torch.cuda.max_memory_allocated() / 1e9))
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        net(imgs, pm, dv); torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=70))
