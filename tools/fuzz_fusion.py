"""Random fusion scenes (sizes, source scale, thresholds, poisoned samples): d3d_consistency_check and the fused
accumulation against the CPU oracle, bit for bit.  Usage: python tools/fuzz_fusion.py [n_cases]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from deep3d_aerial_amd import fuse, synthetic as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(123)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
bad = 0
for case in range(n):
    h, w = int(rng.integers(8, 300)), int(rng.integers(8, 400))
    scale = float(rng.choice([0.5, 0.77, 1.0, 1.0, 1.3]))
    nsrc = int(rng.integers(1, 4))
    ref, srcs = S.make_fusion_scene(h, w, nsrc, seed=1000 + case, noise=float(rng.choice([0.0, 0.004, 0.02])), src_scale=scale)
    thr = (float(rng.choice([0.5, 1.0, 2.0])), float(rng.choice([0.005, 0.01, 0.05])), float(rng.choice([5.0, 10.0, 90.0])),
           float(rng.choice([0.0, 0.2, 0.6])))
    if case % 3 == 0:  # poison
        for arr in [ref["depth"]] + [s["depth"] for s in srcs]:
            idx = rng.integers(0, arr.size, max(1, arr.size // 50))
            arr.reshape(-1)[idx] = rng.choice(np.array([np.nan, np.inf, -1.0, 0.0, 1e30, 1e-30], np.float32), idx.size)
    chk = fuse.ConsistencyChecker(*thr)
    vf = fuse.ViewFusion(chk, dev(ref["depth"]), dev(ref["normal"]), ref["K"], ref["E"], dev(ref["confidence"]), 1)
    xyz, conf, cnt, _ = oracle.fusion.fusion_ref_init(ref["depth"], ref["normal"], ref["K"], ref["E"])
    ok = True
    for i, s in enumerate(srcs):
        with np.errstate(all="ignore"):
            want = oracle.fusion.consistency_check(ref["depth"], ref["normal"], ref["K"], ref["E"], s["depth"], s["normal"],
                                                   s["K"], s["E"], ref["confidence"], *thr)
            oracle.fusion.fusion_accumulate(want[0], want[3], want[4], 2 + i, cnt, xyz, conf)
        got = chk.check(dev(ref["depth"]), dev(ref["normal"]), ref["K"], ref["E"], dev(s["depth"]), dev(s["normal"]), s["K"],
                        s["E"], dev(ref["confidence"]))
        filt = vf.add_source(dev(s["depth"]), dev(s["normal"]), s["K"], s["E"], 2 + i)
        ok &= all(np.array_equal(g.cpu().numpy(), w_, equal_nan=True) for g, w_ in zip(got, want))
        ok &= np.array_equal(filt.cpu().numpy(), want[2], equal_nan=True)
    ok &= np.array_equal(vf.geo_mask_sum.cpu().numpy(), cnt) and np.array_equal(vf.all_xyz_world.cpu().numpy(), xyz, equal_nan=True)
    ok &= np.array_equal(vf.xyz_confidence.cpu().numpy(), conf, equal_nan=True)
    if not ok:
        bad += 1
        print("MISMATCH case %d: %dx%d scale %.2f nsrc %d thr %s" % (case, h, w, scale, nsrc, thr), flush=True)
print("%d cases, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
