#!/usr/bin/env python3
"""Golden vectors for SURVEY.md 8(a) row a2, made by RUNNING THE REFERENCE's homo_warping_double
(mvs/mvs_cas/models/module.py:560-601) on CPU with float64 projection matrices (the only kind it accepts):

    python tests/golden/make_golden_warp_double.py

The cameras carry large world translations (aerial blocks: coordinates of 1e5..1e6 m), the situation the function exists
for: there the fp32 chain of homo_warping_float moves samples by whole pixels while the fp64 chain does not.  The largest
difference to the same inputs through homo_warping_float (fp32 matrices) is stored per case, to show the two differ.
Data only."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("D3D_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "mvs", "mvs_cas"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from deep3d_aerial_amd import synthetic as S  # noqa: E402
from models import module as RM  # noqa: E402  (the reference)


def main():
    rng = np.random.default_rng(560)
    out, n = {}, 0
    for (C, h, w, D, offset) in [(4, 12, 10, 3, 0.0), (8, 24, 20, 5, 3.0e5), (8, 24, 20, 4, 2.0e6), (3, 17, 29, 2, 5.0e4)]:
        proj, dv = S.make_scene(3, h, w, 8, sweep_px=5.0, seed=60 + n)
        P = proj.astype(np.float64)
        # move the whole scene far from the origin: world point X -> X + t0 changes every projection's last column
        t0 = np.array([offset, -0.7 * offset, 0.01 * offset])
        for v in range(P.shape[0]):
            P[v, :3, 3] = P[v, :3, 3] - P[v, :3, :3] @ t0
        src = rng.standard_normal((C, h, w), dtype=np.float32)
        for kind in ("vec", "map"):
            if kind == "vec":
                depth = S.uniform_depths(dv, max(D, 2))[:D].copy()
            else:
                depth = (600 + 120 * rng.standard_normal((D, h, w))).astype(np.float32)
            for vi in (1, 2):
                dt = torch.from_numpy(depth)[None]
                y64 = RM.homo_warping_double(torch.from_numpy(src)[None], torch.from_numpy(P[vi])[None],
                                             torch.from_numpy(P[0])[None], dt)[0].numpy()
                y32 = RM.homo_warping_float(torch.from_numpy(src)[None], torch.from_numpy(P[vi].astype(np.float32))[None],
                                            torch.from_numpy(P[0].astype(np.float32))[None], dt)[0].numpy()
                k = "c%d_" % n
                out.update({k + "src": src, k + "ref_proj": P[0], k + "src_proj": P[vi], k + "depth": depth,
                            k + "out": y64,
                            k + "float_chain_maxdiff": np.array(np.abs(y64 - y32).max())})
                print(n, (C, h, w, D), kind, "offset %.0e" % offset, "max |double - float chain| = %.3g" % np.abs(y64 - y32).max())
                n += 1
    out["n_cases"] = np.array(n)
    path = os.path.join(HERE, "ops_warp_double.npz")
    np.savez_compressed(path, **out)
    print("wrote %s %.1f KiB" % (os.path.basename(path), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
