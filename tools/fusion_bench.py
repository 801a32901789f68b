"""Times row N1 (consistency check / fused accumulation) at the 2752x1856 map size: HIP events around repeated
launches on the current stream, algorithmic bytes / time against the 8 TB/s HBM roofline, and the CPU oracle
(one core) on a 1/16 crop for scale.  Usage: python tools/fusion_bench.py [--h 1856 --w 2752 --pairs 10]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import fuse, synthetic as S  # noqa: E402


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=1856)
    ap.add_argument("--w", type=int, default=2752)
    ap.add_argument("--pairs", type=int, default=10)
    ap.add_argument("--cpu", type=int, default=1)
    a = ap.parse_args()
    ref, srcs = S.make_fusion_scene(a.h, a.w, 2, seed=1)
    dev = lambda x: torch.from_numpy(x).cuda()
    d, n, c = dev(ref["depth"]), dev(ref["normal"]), dev(ref["confidence"])
    sd, sn = dev(srcs[0]["depth"]), dev(srcs[0]["normal"])
    chk = fuse.ConsistencyChecker(1.0, 0.01, 10.0, 0.2)
    px = a.h * a.w
    # plain check: reads 20 B (depth, normal, confidence) + 16 B gathered (depth, normal), writes 1 + 4 + 12 + 12 B
    # and the source-map copy (8 B); fused: reads 36 B, read-modify-write of count, xyz, confidence (2 x 20 B),
    # writes vis 4 B + the source-map copy 8 B
    t_check = timeit(lambda: chk.check(d, n, ref["K"], ref["E"], sd, sn, srcs[0]["K"], srcs[0]["E"], c), a.pairs)
    vf = fuse.ViewFusion(chk, d, n, ref["K"], ref["E"], c, 1)
    t_fused = timeit(lambda: vf.add_source(sd, sn, srcs[0]["K"], srcs[0]["E"], 2), a.pairs)
    vf.vis_infos = vf.vis_infos[:1]
    t_nofilter = timeit(lambda: vf.add_source(sd, sn, srcs[0]["K"], srcs[0]["E"], 2, filter_source=False), a.pairs)
    for name, t, b in (("check (5 outputs)", t_check, 73), ("fused accumulate + filtered source", t_fused, 88),
                       ("fused accumulate", t_nofilter, 80)):
        print("%-36s %8.1f us  %6.2f Gpixel/s  %5.2f TB/s algorithmic (%d B/pixel) = %4.1f %% of 8 TB/s"
              % (name, t * 1e6, px / t * 1e-9, px * b / t * 1e-12, b, px * b / t / 8e12 * 100))
    if a.cpu:
        import oracle
        hh, ww = a.h // 4, a.w // 4
        r2, s2 = S.make_fusion_scene(hh, ww, 1, seed=1)
        t0 = time.time()
        oracle.fusion.consistency_check(r2["depth"], r2["normal"], r2["K"], r2["E"], s2[0]["depth"], s2[0]["normal"],
                                        s2[0]["K"], s2[0]["E"], r2["confidence"], 1.0, 0.01, 10.0, 0.2)
        t = time.time() - t0
        print("CPU oracle (1 core, %dx%d crop): %.1f ms = %.3f Gpixel/s" % (hh, ww, t * 1e3, hh * ww / t * 1e-9))


if __name__ == "__main__":
    main()
