"""Seeded synthetic plane-sweep scenes (cameras, depth range) for tests and bench.py.

The reference ships neither data nor checkpoints, so every parity case
and the benchmark run on inputs built here.  Projection convention follows the
reference dataset item builder (datasets/cas_normal_eval.py:138-162): a 4x4 matrix
whose top three rows are K @ [R|t] (world -> pixel) and whose last row is [0,0,0,1].
numpy only -- no torch, no GPU.
"""
import numpy as np


def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def make_scene(n_views, h, w, num_planes, sweep_px=None, depth_min=400.0, depth_max=800.0, focal=None,
               yaw_deg=1.0, seed=0, scale=1.0):
    """Cameras for one reference view + (n_views-1) sources at feature resolution h x w.

    The reference camera sits at the origin looking down +z.  Source i is displaced by a
    baseline chosen so that its disparity changes by about `sweep_px` pixels across
    [depth_min, depth_max] (default num_planes/2), converged on the
    mid-depth plane so the sweep is centred in the frame, and rolled by a small yaw.

    Returns proj [n_views,4,4] float32 and depth_values [2] float32 = (depth_min, depth_max).
    `scale` multiplies the first two rows (the reference's stage scaling, e.g. 0.5, 0.25).
    """
    rng = np.random.default_rng(seed)
    if sweep_px is None:
        sweep_px = num_planes / 2.0
    if focal is None:
        focal = 1.2 * max(h, w)
    K = np.array([[focal, 0, (w - 1) / 2.0], [0, focal, (h - 1) / 2.0], [0, 0, 1.0]])
    d_mid = 2.0 / (1.0 / depth_min + 1.0 / depth_max)
    base = sweep_px / (focal * (1.0 / depth_min - 1.0 / depth_max))
    dirs = [(1, 0), (-1, 0), (0, 1), (0, -1), (0.7, 0.7), (-0.7, 0.7), (0.7, -0.7), (-0.7, -0.7)]
    projs = []
    for v in range(n_views):
        if v == 0:
            R, C = np.eye(3), np.zeros(3)
        else:
            dx, dy = dirs[(v - 1) % len(dirs)]
            b = base * (1.0 + 0.15 * rng.standard_normal())
            C = np.array([dx * b, dy * b, 0.02 * b * rng.standard_normal()])
            # converge on the point (0,0,d_mid): rotate so that it projects to the principal point
            ry = np.arctan2(C[0], d_mid)
            rx = -np.arctan2(C[1], d_mid)
            rz = np.deg2rad(yaw_deg) * rng.standard_normal()
            R = _rot(rx, ry, rz)
        t = -R @ C
        P = np.eye(4)
        P[:3, :3] = K @ R
        P[:3, 3] = K @ t
        P[:2, :] *= scale
        projs.append(P)
    return np.stack(projs).astype(np.float32), np.array([depth_min, depth_max], np.float32)


def model_inputs(V, H, W, num_depth, seed):
    """Seeded inputs of a whole Infer_* forward: imgs [1,V,3,H,W], proj_matrices {stageN: [1,V,4,4]}, depth_values [1,2].
    The model fixtures (tests/golden/make_golden.py) are the reference's outputs on exactly these; the large fixtures
    store only the outputs and the tests regenerate the inputs from (V, H, W, num_depth, seed)."""
    rng = np.random.default_rng(seed)
    imgs = rng.standard_normal((1, V, 3, H, W), dtype=np.float32)
    # smooth a little so features are not pure noise
    imgs = (imgs + np.roll(imgs, 1, -1) + np.roll(imgs, 1, -2)) / 1.7
    proj_full, dv = make_scene(V, H, W, num_depth, sweep_px=24.0, seed=seed, yaw_deg=2.0)
    pm = {}
    for name, sc in (("stage1", 0.25), ("stage2", 0.5), ("stage3", 1.0)):
        p = proj_full.copy()
        p[:, :2, :] = proj_full[:, :2, :] * np.float32(sc)
        pm[name] = p[None]
    return imgs.astype(np.float32), pm, dv[None]


def uniform_depths(depth_values, num_planes):
    """linspace(min, max, D) as float32 -- the stage-1 hypothesis set (module.py:637-642)."""
    lo, hi = np.float32(depth_values[0]), np.float32(depth_values[1])
    step = np.float32((hi - lo) / np.float32(num_planes - 1))
    return (lo + np.arange(num_planes, dtype=np.float32) * step).astype(np.float32)


def make_features(n_views, channels, h, w, seed=0, smooth=False):
    """N(0,1) features [V,C,h,w] float32 (optionally box-smoothed so they resemble CNN features)."""
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((n_views, channels, h, w), dtype=np.float32)
    if smooth:
        f = (f + np.roll(f, 1, -1) + np.roll(f, 1, -2) + np.roll(np.roll(f, 1, -1), 1, -2)) * 0.5
    return f


def in_frame_fraction(proj, depths, h, w, stride=8):
    """Fraction of (src, d, y, x) samples whose bilinear footprint touches the source image."""
    proj = proj.astype(np.float64)
    ys, xs = np.meshgrid(np.arange(0, h, stride), np.arange(0, w, stride), indexing="ij")
    pix = np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)])
    tot, hit = 0, 0
    for i in range(1, proj.shape[0]):
        M = proj[i] @ np.linalg.inv(proj[0])
        for d in np.asarray(depths, np.float64)[::max(1, len(depths) // 16)]:
            p = M[:3, :3] @ pix * d + M[:3, 3:4]
            u, v = p[0] / p[2], p[1] / p[2]
            ok = (u > -1) & (u < w) & (v > -1) & (v < h) & (p[2] > 0)
            tot += ok.size
            hit += int(ok.sum())
    return hit / max(tot, 1)


def fill_state_dict_(state_dict, seed):
    """Deterministic, reference-independent weights for any module's state_dict (in place).

    Keys are visited in sorted order and filled from numpy's PCG64 stream, so the golden
    generator (which fills the reference's modules) and the tests (which fill this
    package's modules with the same keys) obtain identical weights without shipping a
    checkpoint.  He-style scaling keeps activations O(1) so softmax outputs are not flat.
    """
    import torch

    rng = np.random.default_rng(seed)
    with torch.no_grad():
        for k in sorted(state_dict.keys()):
            t = state_dict[k]
            shape = tuple(t.shape)
            if k.endswith("num_batches_tracked"):
                continue
            if k.endswith("running_var"):
                a = rng.uniform(0.5, 1.5, shape)
            elif k.endswith("running_mean"):
                a = 0.1 * rng.standard_normal(shape)
            elif t.dim() == 1 and k.endswith("weight"):
                a = rng.uniform(0.5, 1.5, shape)
            elif t.dim() == 1:
                a = 0.1 * rng.standard_normal(shape)
            else:
                fan_in = int(np.prod(shape[1:]))
                a = rng.standard_normal(shape) * np.sqrt(2.0 / max(fan_in, 1))
            t.copy_(torch.from_numpy(np.asarray(a, dtype=np.float32)).reshape(shape))
    return state_dict


LOGIT_LAYER_SUFFIXES = (".prob.weight", ".prob.bias", ".upconv2d.weight", ".upconv2d.bias")


def sharpen_state_dict_(state_dict, gain):
    """Multiplies the last (logit) layer of every regulariser by `gain`, in place: cost_regularization.N.prob of the 3-D UNets
    (cas_mvsnet.py:95, ucsnet.py:70), reg_fuse.upconv2d / reg.prob of AdaMVS (adamvs.py:412-413, 216) and
    cost_regularization.N.upconv2d of RED-Net (msrednet.py:352).  With the He-scaled weights of fill_state_dict_ the softmax /
    exp-sum over the depth planes is nearly flat (confidence ~ 4/D), and a flat distribution regresses to the middle of the
    hypothesis range whatever the regulariser computed; with the logits scaled up the distribution is peaked and the regressed
    depth follows the arg-max plane -- the "peaked" model fixtures (tests/golden/make_golden.py) use this on the reference's
    modules and the tests on this package's, which have the same keys."""
    import torch

    n = 0
    with torch.no_grad():
        for k, t in state_dict.items():
            if k.endswith(LOGIT_LAYER_SUFFIXES):
                t.mul_(float(gain))
                n += 1
    return n


def make_fusion_scene(h, w, n_src=3, seed=0, noise=0.004, src_scale=1.0):
    """A reference view and n_src source views of one tilted ground plane, as the fusion step reads them
    (fuse/fusion_3d_normal.py:425-505): per view a depth map [h,w], a camera-space normal map [h,w,3], K [3,3] and
    E = Tcw [4,4] (float32), plus the reference confidence map.  Source depths carry relative noise of `noise`
    (about half the default 1 % depth threshold), normals a few degrees of noise, parts of the maps are holes
    (depth 0) and low confidence, and the source cameras are offset sideways so part of every reprojection leaves
    the source image.  Source maps are [round(h*src_scale), round(w*src_scale)].
    Returns (ref, [src...]) with ref/src dicts: depth, normal, K, E (+ ref["confidence"])."""
    rng = np.random.default_rng(seed)
    n_w = np.array([0.06, -0.04, -1.0])
    n_w /= np.linalg.norm(n_w)
    c_w = n_w @ np.array([0.0, 0.0, 600.0])

    def view(hh, ww, f, R, C, rel_noise):
        K = np.array([[f, 0, (ww - 1) / 2.0], [0, f, (hh - 1) / 2.0], [0, 0, 1]], np.float64)
        t = -R @ C
        ys, xs = np.mgrid[0:hh, 0:ww]
        rays = np.linalg.inv(K) @ np.stack([xs.ravel(), ys.ravel(), np.ones(hh * ww)])
        nr = n_w @ R.T
        d = ((c_w + nr @ t) / (nr @ rays)).reshape(hh, ww)
        d = d * (1.0 + rel_noise * rng.standard_normal((hh, ww)))
        n_cam = R @ n_w
        normal = n_cam[None, None, :] + 0.03 * rng.standard_normal((hh, ww, 3))
        E = np.eye(4)
        E[:3, :3] = R
        E[:3, 3] = t
        return dict(depth=d.astype(np.float32), normal=normal.astype(np.float32), K=K.astype(np.float32),
                    E=E.astype(np.float32))

    ref = view(h, w, 1.4 * w, _rot(0.01, -0.02, 0.015), np.zeros(3), 0.0)
    ref["confidence"] = rng.uniform(0.0, 1.0, (h, w)).astype(np.float32)
    ref["depth"][rng.uniform(size=(h, w)) < 0.02] = 0.0  # holes
    hs, ws = int(round(h * src_scale)), int(round(w * src_scale))
    srcs = []
    for i in range(n_src):
        a = 2.0 * np.pi * i / max(n_src, 1)
        C = np.array([90.0 * np.cos(a), 60.0 * np.sin(a), 10.0 * (i - 1)])
        s = view(hs, ws, 1.4 * ws, _rot(0.02 * np.sin(a), -0.03 * np.cos(a), 0.02 * (i - 1)), C, noise)
        s["depth"][rng.uniform(size=(hs, ws)) < 0.02] = 0.0
        srcs.append(s)
    return ref, srcs
