#!/usr/bin/env python3
"""Headline benchmark: fused homography-warp + variance cost volume (BASELINE.json config 2).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one reference view: 5 views x 384 depth planes over 32 x 688 x 464 fp32 feature
maps (1/4-resolution features of a 2752 x 1856 image) -> the [32,384,688,464] variance
volume, through the C-ABI kernel d3d_variance_volume.  Inputs are synthetic (seeded N(0,1)
features, converging cameras, deep3d_aerial_amd/synthetic.py) and resident in HBM before the
timed region.  With N > 1 every rank sweeps its own reference views (independent work, no
collective on the data path): weak scaling, value = all ranks' voxels / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline      -- algorithmic HBM bytes per launch / mean kernel time from HIP events
  cpu_baseline  -- the CPU oracle (oracle/planesweep_oracle.c, OpenMP) timed on a bounded
                   sample of the same workload on this box's host cores (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from deep3d_aerial_amd import ops, synthetic as S  # noqa: E402

V, C, D, H_FEAT, W_FEAT = 5, 32, 384, 688, 464
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes():
    """read V feature maps once + write the fp32 volume once."""
    reads = V * C * H_FEAT * W_FEAT * 4
    writes = C * D * H_FEAT * W_FEAT * 4
    return reads + writes


def host_threads():
    """Host cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(proj34_host, feats_host, depth_host):
    """Times the CPU oracle on a depth sub-range of the same workload (bounded to ~10-20 s)."""
    import oracle

    oracle.build()
    cores = host_threads()
    oracle.set_num_threads(cores)
    probe = 4
    t0 = time.perf_counter()
    oracle.variance_volume(feats_host[0], feats_host[1:], proj34_host, depth_host[:probe])
    dt = time.perf_counter() - t0
    planes = int(min(D, max(probe, probe * 12.0 / max(dt, 1e-3))))
    start = (D - planes) // 2
    t0 = time.perf_counter()
    oracle.variance_volume(feats_host[0], feats_host[1:], proj34_host, depth_host[start:start + planes])
    dt = time.perf_counter() - t0
    vox = planes * H_FEAT * W_FEAT
    return {"value": round(vox / dt / 1e6, 3), "unit": "Mvoxels/s", "cores": oracle.num_threads(), "kind": "port",
            "sample": "%d of %d depth planes (planes %d..%d) of the same 5-view 32x688x464 workload, %.1f s"
                      % (planes, D, start, start + planes - 1, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    # every rank builds its own reference view (seed = rank): independent work items
    proj, dv = S.make_scene(V, H_FEAT, W_FEAT, D, seed=rank)
    feats_host = S.make_features(V, C, H_FEAT, W_FEAT, seed=rank)
    depth_host = S.uniform_depths(dv, D)
    in_frame = S.in_frame_fraction(proj, depth_host, H_FEAT, W_FEAT)
    feats = [torch.from_numpy(f).cuda() for f in feats_host]
    depth = torch.from_numpy(depth_host).cuda()
    p34 = ops.compose_projections(torch.from_numpy(proj).cuda())
    out = torch.empty((C, D, H_FEAT, W_FEAT), dtype=torch.float32, device="cuda")

    def step():
        ops.variance_volume(feats, p34, depth, out=out)

    for _ in range(args.warmup):
        step()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    stops = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        starts[i].record()  # same stream as the kernel (torch's current stream is passed to the C ABI)
        step()
        stops[i].record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kern_ms = float(np.mean([s.elapsed_time(e) for s, e in zip(starts, stops)]))
    voxels = D * H_FEAT * W_FEAT
    value = world * args.steps * voxels / elapsed / 1e6
    achieved = algorithmic_bytes() / (kern_ms * 1e-3) / 1e9

    if rank == 0:
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "cost-volume Mvoxels/s (5v x 384D)",
            "value": round(value, 1),
            "unit": "Mvoxels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "config 2: fused homography-warp + variance, 5 views x 384 planes, "
                                   "features 32x688x464 fp32 (2752x1856 image at 1/4 res), one reference view per "
                                   "step per GPU", "voxels_per_step_per_gpu": voxels,
                       "in_frame_fraction": round(in_frame, 4), "path": os.environ.get("D3D_FORCE_PATH", "auto")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel_ms": round(kern_ms, 4), "algorithmic_bytes": algorithmic_bytes()},
        }
        if world == 1 and not args.no_cpu_baseline:
            p34_host = p34.cpu().numpy().reshape(-1, 3, 4)
            line["cpu_baseline"] = cpu_baseline(p34_host, feats_host, depth_host)
        print(json.dumps(line), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
