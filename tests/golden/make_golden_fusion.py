#!/usr/bin/env python3
"""Golden vectors for SURVEY.md §8f row N1 (geometric consistency check), made by RUNNING THE REFERENCE's
ConsistencyChecker.check_cupy (fuse/consistency_check_n.py:29-138) in the build container:

    python tests/golden/make_golden_fusion.py

The reference module does `import cupy as cp`; CuPy is not installed here.  CuPy's array API is NumPy's, so for the
duration of this script the name `cupy` is bound to a module object that forwards every attribute to NumPy (plus
`asnumpy`, and `int`, which NumPy 2 no longer has).  Nothing of the reference is copied: the module is imported in
place from /root/reference (sys.dont_write_bytecode keeps the tree clean).

Limits of this pinning, stated in oracle/fusion_oracle.c and DESIGN.md: NumPy raises on out-of-range indices
where CuPy wraps around, so the scenes here keep every reprojection inside the source image (a source camera with
a wider field of view; for the zero-depth holes of scene B a source camera that sees the reference camera's
centre).  The per-view accumulation (fusion_3d_normal.py:513-527) is a method that reads its inputs from files
and cannot be run in isolation; those six array statements are checked by construction in the tests.

The .npz files hold data only: the seeded inputs and the reference's five outputs.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("D3D_REFERENCE", "/root/reference")

import numpy as np  # noqa: E402


class _NumpyAsCupy(types.ModuleType):
    int = int

    @staticmethod
    def asnumpy(a):
        return np.asarray(a)

    def __getattr__(self, name):
        return getattr(np, name)


sys.modules["cupy"] = _NumpyAsCupy("cupy")
sys.path.insert(0, os.path.join(REF, "fuse"))
import consistency_check_n as RCN  # noqa: E402


def rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def camera(f, h, w, R, C):
    """K [3,3], E = Tcw [4,4] (x_cam = R (X - C)), float32 as read_camera_parameters returns them."""
    K = np.array([[f, 0, (w - 1) / 2.0], [0, f, (h - 1) / 2.0], [0, 0, 1]], np.float32)
    E = np.eye(4)
    E[:3, :3] = R
    E[:3, 3] = -R @ C
    return K, E.astype(np.float32)


def render_plane(K, E, h, w, n_w, c_w):
    """Depth and camera-space normal maps of the world plane n_w . X = c_w seen by camera (K, E)."""
    K, E = K.astype(np.float64), E.astype(np.float64)
    R, t = E[:3, :3], E[:3, 3]
    ys, xs = np.mgrid[0:h, 0:w]
    rays = np.linalg.inv(K) @ np.stack([xs.ravel(), ys.ravel(), np.ones(h * w)])  # camera rays, z = 1
    # X = R^T (d * ray - t);  n.X = c  ->  d = (c + n.R^T t) / (n.R^T ray)
    nr = n_w @ R.T
    d = (c_w + nr @ t) / (nr @ rays)
    n_cam = R @ n_w
    n_cam = n_cam / np.linalg.norm(n_cam)
    normal = np.broadcast_to(n_cam.astype(np.float32), (h, w, 3)).copy()
    return d.reshape(h, w).astype(np.float32), normal


def scene(tag, seed):
    rng = np.random.default_rng(seed)
    h, w, hs, ws = 40, 56, 46, 62
    n_w = np.array([0.08, -0.05, -1.0])
    n_w /= np.linalg.norm(n_w)
    c_w = n_w @ np.array([0.0, 0.0, 60.0])
    Kr, Er = camera(1.3 * w, h, w, rot(0.02, -0.03, 0.01), np.array([0.0, 0.0, 0.0]))
    if tag == "lateral":
        Ks, Es = camera(0.9 * ws, hs, ws, rot(-0.03, 0.05, -0.02), np.array([4.0, -1.5, 0.5]))
    else:  # "forward": the source camera sits behind the reference camera and sees its centre
        Ks, Es = camera(0.8 * ws, hs, ws, rot(0.01, 0.02, 0.03), np.array([0.6, -0.4, -9.0]))
    d_ref, n_ref = render_plane(Kr, Er, h, w, n_w, c_w)
    d_src, n_src = render_plane(Ks, Es, hs, ws, n_w, c_w)
    # smooth relief on the reference depth below the depth threshold, so reprojection errors are not all zero
    yy, xx = np.mgrid[0:h, 0:w]
    d_ref = (d_ref * (1.0 + 0.002 * np.sin(xx / 5.0) * np.cos(yy / 7.0))).astype(np.float32)
    # inconsistent regions: wrong source depth, tilted source normals, low confidence, zero-depth holes
    d_src[5:14, 8:30] *= np.float32(1.03)
    tilt = rot(0.5, 0.0, 0.0).astype(np.float32)
    n_src[20:30, 30:50] = n_src[20:30, 30:50] @ tilt.T
    n_ref = (n_ref + 0.02 * rng.standard_normal(n_ref.shape)).astype(np.float32)
    n_src = (n_src + 0.02 * rng.standard_normal(n_src.shape)).astype(np.float32)
    prob = rng.uniform(0.25, 1.0, (h, w)).astype(np.float32)
    prob[30:36, 4:20] = 0.1
    if tag == "forward":
        d_ref[10:16, 36:48] = 0.0
        d_ref[33, :] = 0.0
    return dict(depth_ref=d_ref, normal_ref=n_ref, K_ref=Kr, E_ref=Er, depth_src=d_src, normal_src=n_src, K_src=Ks,
                E_src=Es, prob_ref=prob)


def main():
    for tag, seed, thr in [("lateral", 9101, (1.0, 0.01, 10.0, 0.2)), ("forward", 9102, (0.75, 0.01, 25.0, 0.3))]:
        s = scene(tag, seed)
        chk = RCN.ConsistencyChecker(thr[0], thr[1], thr[2], thr[3], implement="cupy")
        with np.errstate(all="ignore"):
            mask, drep, dsrc, xyz, ang = chk.check(s["depth_ref"].copy(), s["normal_ref"].copy(), s["K_ref"].copy(),
                                                   s["E_ref"].copy(), s["depth_src"].copy(), s["normal_src"].copy(),
                                                   s["K_src"].copy(), s["E_src"].copy(), s["prob_ref"].copy())
        path = os.path.join(HERE, "fusion_pair_%s.npz" % tag)
        np.savez_compressed(path, thresholds=np.array(thr, np.float64), out_mask=mask, out_depth_reprojected=drep,
                            out_depth_src=dsrc, out_xyz_world_src=xyz, out_angle_conf=ang, **s)
        print("wrote %s  %.1f KiB  consistent %.3f  zeroed src samples %d  dtypes %s" % (
            os.path.basename(path), os.path.getsize(path) / 1024, mask.mean(), int((dsrc == 0).sum()),
            [a.dtype.name for a in (mask, drep, dsrc, xyz, ang)]))


if __name__ == "__main__":
    main()
