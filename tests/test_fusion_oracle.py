"""Row N1, CPU side: pins oracle/fusion_oracle.c against the outputs of the reference's
ConsistencyChecker.check_cupy (tests/golden/fusion_pair_*.npz, made by tests/golden/make_golden_fusion.py) and checks
the accumulator statements of fuse/fusion_3d_normal.py:513-527 by construction."""
import numpy as np
import pytest

from conftest import load_golden
from deep3d_aerial_amd import synthetic as S

# The geometric chain is float64 with float32 casts at the reference's own rounding points: identical results.
# The cosine of the normals is a float32 matmul + reduction whose summation order belongs to the BLAS: 2 ulp.
COS_ABS = 3e-7


@pytest.mark.parametrize("tag", ["lateral", "forward"])
def test_consistency_check_matches_reference(oracle, tag):
    g = load_golden("fusion_pair_" + tag)
    t = g["thresholds"]
    mask, drep, dsrc, xyz, ang = oracle.fusion.consistency_check(
        g["depth_ref"], g["normal_ref"], g["K_ref"], g["E_ref"], g["depth_src"], g["normal_src"], g["K_src"],
        g["E_src"], g["prob_ref"], t[0], t[1], t[2], t[3])
    assert 0.2 < g["out_mask"].mean() < 0.9  # the fixture exercises both outcomes
    assert np.array_equal(mask, g["out_mask"])
    assert np.array_equal(drep, g["out_depth_reprojected"])
    assert np.array_equal(dsrc, g["out_depth_src"])
    assert np.array_equal(xyz, g["out_xyz_world_src"])
    assert np.abs(ang - g["out_angle_conf"]).max() <= COS_ABS
    assert np.array_equal(ang == 0, g["out_angle_conf"] == 0)
    if tag == "forward":  # zero-depth holes are never consistent and their outputs are zero
        holes = g["depth_ref"] == 0
        assert holes.any() and not mask[holes].any() and not xyz[:, holes].any()


def test_scatter_uses_the_sampled_pixels(oracle):
    """consistency_check_n.py:123-126: exactly the source samples of consistent pixels are zeroed."""
    g = load_golden("fusion_pair_lateral")
    t = g["thresholds"]
    mask, _, dsrc, _, _ = oracle.fusion.consistency_check(
        g["depth_ref"], g["normal_ref"], g["K_ref"], g["E_ref"], g["depth_src"], g["normal_src"], g["K_src"],
        g["E_src"], g["prob_ref"], t[0], t[1], t[2], t[3])
    changed = dsrc != g["depth_src"]
    assert changed.any() and (dsrc[changed] == 0).all()
    assert changed.sum() <= mask.sum()


def test_out_of_range_samples_wrap(oracle):
    """Reprojections that leave the source image index it modulo its size (CuPy's integer-array indexing; NumPy
    would raise, so the golden fixtures cannot cover it).  Property: tiling the source maps 2x2 turns wrapped
    indices into in-range ones that address the same values, and the back-projection uses the unwrapped integer
    index, so every per-pixel output is unchanged -- true for modulo wrapping only (clamping would fail)."""
    ref, srcs = S.make_fusion_scene(48, 64, 1, seed=3)
    s = srcs[0]
    args = (ref["K"], ref["E"]), (s["K"], s["E"], ref["confidence"], 1.0, 0.01, 10.0, 0.2)
    a = oracle.fusion.consistency_check(ref["depth"], ref["normal"], *args[0], s["depth"], s["normal"], *args[1])
    b = oracle.fusion.consistency_check(ref["depth"], ref["normal"], *args[0], np.tile(s["depth"], (2, 2)),
                                        np.tile(s["normal"], (2, 2, 1)), *args[1])
    for u, v in zip((a[0], a[1], a[3], a[4]), (b[0], b[1], b[3], b[4])):
        assert np.array_equal(u, v)
    # the scene does leave the source image (projection of the reference pixels, float64 numpy)
    ys, xs = np.mgrid[0:48, 0:64]
    pts = np.linalg.inv(ref["K"]).astype(np.float64) @ (np.stack([xs.ravel(), ys.ravel(), np.ones(48 * 64)]) * ref["depth"].ravel())
    M = s["E"].astype(np.float64) @ np.linalg.inv(ref["E"].astype(np.float64))
    q = s["K"].astype(np.float64) @ (M[:3, :3] @ pts + M[:3, 3:4])
    with np.errstate(all="ignore"):
        u = q[0] / q[2]
    assert (u[ref["depth"].ravel() > 0] > 64.5).any() or (u[ref["depth"].ravel() > 0] < -0.5).any()


def test_accumulators_by_construction(oracle):
    """fusion_3d_normal.py:452-474, 513-518, 522-527 written out with numpy on the oracle's check outputs."""
    ref, srcs = S.make_fusion_scene(40, 56, 3, seed=5)
    H, W = ref["depth"].shape
    xyz, conf, cnt, nw = oracle.fusion.fusion_ref_init(ref["depth"], ref["normal"], ref["K"], ref["E"])
    # reference-view world points: inv(E) @ [inv(K) @ ([x,y,1] * d); 1]
    ys, xs = np.mgrid[0:H, 0:W]
    grid = np.vstack((xs.reshape(-1), ys.reshape(-1), np.ones(H * W, np.int64))) * ref["depth"].reshape(-1)
    cam_pts = np.matmul(np.linalg.inv(ref["K"]), grid)
    want = np.matmul(np.linalg.inv(ref["E"]), np.vstack((cam_pts, np.ones(H * W))))[:3].reshape(3, H, W)
    assert np.abs(xyz - want.astype(np.float32)).max() <= 1e-4 * np.abs(want).max()
    assert (conf == 1).all() and (cnt == 1).all()
    assert np.abs(np.linalg.norm(nw, axis=-1) - 1).max() < 1e-6
    e_xyz, e_conf, e_cnt = xyz.copy(), np.repeat(conf[None], 3, 0), cnt.copy()
    for i, s in enumerate(srcs):
        m, _, _, pts, ang = oracle.fusion.consistency_check(ref["depth"], ref["normal"], ref["K"], ref["E"], s["depth"],
                                                            s["normal"], s["K"], s["E"], ref["confidence"], 1.0, 0.01,
                                                            10.0, 0.2)
        vis = oracle.fusion.fusion_accumulate(m, pts, ang, 7 + i, cnt, xyz, conf)
        e_cnt += m.astype(np.int32)
        e_xyz += (ang * pts).astype(np.float32)
        e_conf += ang
        assert np.array_equal(vis, m.astype(np.int32) * (7 + i))
    assert np.array_equal(cnt, e_cnt) and np.array_equal(xyz, e_xyz) and np.array_equal(conf, e_conf[0])
    avg, fm = oracle.fusion.fusion_finalize(xyz, conf, cnt, 3)
    assert np.array_equal(avg, (e_xyz / e_conf).astype(np.float32))
    assert np.array_equal(fm, e_cnt >= 3) and 0 < fm.mean() < 1


def _points_scene(h, w, n_vis, seed):
    rng = np.random.default_rng(seed)
    avg = (rng.standard_normal((3, h, w)) * 40 + np.array([500.0, -200.0, 30.0])[:, None, None]).astype(np.float32)
    avg[0, rng.random((h, w)) < 0.01] = np.nan                       # a failed average (0/0) must drop out at the range test
    mask = rng.random((h, w)) < 0.35
    vis = [np.full((h, w), 3, np.int32)] + [(rng.random((h, w)) < 0.6).astype(np.int32) * int(v) for v in rng.permutation(np.arange(4, 4 + n_vis - 1))]
    color = (rng.integers(0, 256, (h, w, 3)).astype(np.float32) / 255.).astype(np.float32)
    normal = rng.standard_normal((h, w, 3)).astype(np.float32)
    return avg, mask, vis, color, normal, [470.0, 545.0, -250.0, -160.0, 0.0, 100.0]


def _points_reference_loop(avg, mask, vis, color, normal, skip_line, scene_range):
    """The statements of fuse/fusion_3d_normal.py:545-570, with NumPy, as the construction check of the oracle."""
    all_vis = np.array([v[mask] for v in vis])
    xyz = avg[np.repeat(mask[None], 3, axis=0)].reshape(3, -1)
    col = (color[mask] * 255).astype(int)
    nrm = normal[mask]
    V, C, N, VW = [], [], [], []
    for i in range(0, xyz.shape[1], skip_line):
        if len(all_vis[0]) > 1:
            p, va = xyz[:, i], all_vis[:, i]
            if scene_range[0] < p[0] < scene_range[1] and scene_range[2] < p[1] < scene_range[3]:
                V.append(p); C.append(col[i]); N.append(nrm[i]); VW.append(sorted((va[va > 0] - 1).tolist()))
    return V, C, N, VW


@pytest.mark.parametrize("skip_line", [1, 2, 3])
def test_fusion_points_by_construction(oracle, skip_line):
    """Row N1 tail: the oracle's vertex extraction equals the reference's boolean-index compaction + per-point loop."""
    from oracle import fusion as F

    avg, mask, vis, color, normal, sr = _points_scene(37, 53, 5, 11 + skip_line)
    xyz, oc, on, ov, onv = F.fusion_points(avg, mask, vis, color, normal, skip_line, sr)
    V, C, N, VW = _points_reference_loop(avg, mask, vis, color, normal, skip_line, sr)
    assert len(V) == xyz.shape[0] and len(V) > 20
    assert np.array_equal(xyz, np.array(V, np.float32)) and np.array_equal(oc, np.array(C, np.int32))
    assert np.array_equal(on, np.array(N, np.float32))
    for i, want in enumerate(VW):
        assert ov[i, :onv[i]].tolist() == want and (ov[i, onv[i]:] == -1).all()
    # a single valid pixel emits nothing (:555)
    one = np.zeros_like(mask); one[3, 4] = True
    assert F.fusion_points(avg, one, vis, color, normal, 1, [-1e9, 1e9, -1e9, 1e9])[0].shape[0] == 0
