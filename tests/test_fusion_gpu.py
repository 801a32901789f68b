"""Row N1, GPU side: csrc/fusion.hip through the C-ABI (deep3d_aerial_amd.fuse -> ctypes) against the
reference's own outputs (tests/golden/fusion_pair_*.npz), against the CPU oracle on seeded scenes, and at the full
2752x1856 map size through properties.

The kernel and the oracle perform the same float64 / float32 operations in the same order with contraction off,
and fp64/fp32 divide and square root are correctly rounded on gfx950, so the comparison is bit-exact."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from deep3d_aerial_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fuse():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from deep3d_aerial_amd import _lib, fuse as _fuse

    _lib.load()
    return _fuse


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("tag", ["lateral", "forward"])
def test_check_matches_reference_outputs(fuse, tag):
    g = load_golden("fusion_pair_" + tag)
    t = g["thresholds"]
    chk = fuse.ConsistencyChecker(t[0], t[1], t[2], t[3])
    mask, drep, dsrc, xyz, ang = chk.check(dev(g["depth_ref"]), dev(g["normal_ref"]), g["K_ref"], g["E_ref"],
                                           dev(g["depth_src"]), dev(g["normal_src"]), g["K_src"], g["E_src"],
                                           dev(g["prob_ref"]))
    assert np.array_equal(host(mask), g["out_mask"])
    assert np.array_equal(host(drep), g["out_depth_reprojected"])
    assert np.array_equal(host(dsrc), g["out_depth_src"])
    assert np.array_equal(host(xyz), g["out_xyz_world_src"])
    assert np.abs(host(ang) - g["out_angle_conf"]).max() <= 3e-7  # float32 BLAS summation order of the reference


@pytest.mark.parametrize("h,w,scale,seed", [(120, 160, 1.0, 1), (97, 131, 1.25, 2), (256, 384, 0.5, 3)])
def test_check_bit_exact_vs_oracle(fuse, oracle, h, w, scale, seed):
    ref, srcs = S.make_fusion_scene(h, w, 2, seed=seed, src_scale=scale)
    chk = fuse.ConsistencyChecker(1.0, 0.01, 10.0, 0.2)
    for s in srcs:
        want = oracle.fusion.consistency_check(ref["depth"], ref["normal"], ref["K"], ref["E"], s["depth"], s["normal"],
                                               s["K"], s["E"], ref["confidence"], 1.0, 0.01, 10.0, 0.2)
        got = chk.check(dev(ref["depth"]), dev(ref["normal"]), ref["K"], ref["E"], dev(s["depth"]), dev(s["normal"]),
                        s["K"], s["E"], dev(ref["confidence"]))
        assert 0.1 < want[0].mean() < 0.9
        for name, a, b in zip(("mask", "depth_reprojected", "depth_src", "xyz_world_src", "angle"), got, want):
            assert np.array_equal(host(a), b), name


def test_view_fusion_matches_oracle_chain(fuse, oracle):
    ref, srcs = S.make_fusion_scene(150, 200, 4, seed=11)
    chk = fuse.ConsistencyChecker(1.0, 0.01, 10.0, 0.2)
    vf = fuse.ViewFusion(chk, dev(ref["depth"]), dev(ref["normal"]), ref["K"], ref["E"], dev(ref["confidence"]), 5)
    xyz, conf, cnt, nw = oracle.fusion.fusion_ref_init(ref["depth"], ref["normal"], ref["K"], ref["E"])
    assert np.array_equal(host(vf.all_xyz_world), xyz) and np.array_equal(host(vf.normal_world), nw)
    for i, s in enumerate(srcs):
        m, _, dso, pts, ang = oracle.fusion.consistency_check(ref["depth"], ref["normal"], ref["K"], ref["E"], s["depth"],
                                                              s["normal"], s["K"], s["E"], ref["confidence"], 1.0, 0.01,
                                                              10.0, 0.2)
        vis = oracle.fusion.fusion_accumulate(m, pts, ang, 20 + i, cnt, xyz, conf)
        filtered = vf.add_source(dev(s["depth"]), dev(s["normal"]), s["K"], s["E"], 20 + i)
        assert np.array_equal(host(filtered), dso)
        assert np.array_equal(host(vf.vis_infos[-1]), vis)
    assert np.array_equal(host(vf.geo_mask_sum), cnt)
    assert np.array_equal(host(vf.all_xyz_world), xyz)
    assert np.array_equal(host(vf.xyz_confidence), conf)
    avg, fm = vf.finalize(3)
    want_avg, want_fm = oracle.fusion.fusion_finalize(xyz, conf, cnt, 3)
    assert np.array_equal(host(avg), want_avg) and np.array_equal(host(fm), want_fm)
    assert 0.05 < want_fm.mean() < 0.95
    assert (host(vf.vis_infos[0]) == 5).all()


def test_full_size_identity_pair(fuse):
    """2752x1856 (BASELINE config 2 map size): a view checked against itself reprojects every pixel onto itself, so
    the mask is exactly (confidence > threshold) & (depth > 0), the reprojected depth equals the depth to float32
    rounding of the round trip, the world points equal the reference-view initialisation, and the filtered source
    map is zero exactly on the mask."""
    H, W = 1856, 2752
    ref, _ = S.make_fusion_scene(H, W, 0, seed=21)
    d, n, c = dev(ref["depth"]), dev(ref["normal"]), dev(ref["confidence"])
    chk = fuse.ConsistencyChecker(1.0, 0.01, 10.0, 0.2)
    mask, drep, dsrc, xyz, ang = chk.check(d, n, ref["K"], ref["E"], d, n, ref["K"], ref["E"], c)
    want = (c > 0.2) & (d > 0)
    assert torch.equal(mask, want)
    assert float((drep[want] / d[want] - 1).abs().max()) < 1e-5
    assert torch.equal(dsrc == 0, want | (d == 0))
    vf = fuse.ViewFusion(chk, d, n, ref["K"], ref["E"], c, 1)
    rel = (xyz - vf.all_xyz_world * want).abs().max() / vf.all_xyz_world.abs().max()
    assert float(rel) < 1e-5
    assert float((ang[0][want] - 1).abs().max()) < 1e-5 and float(ang[0][~want].abs().max()) == 0
    # fused accumulation of the same pair: count 2 on the mask, confidence 1 + cos, average unchanged
    vf.add_source(d, n, ref["K"], ref["E"], 2, filter_source=False)
    avg, fm = vf.finalize(2)
    assert torch.equal(fm, want)
    assert float((avg - xyz)[:, want].abs().max() / xyz.abs().max()) < 1e-5


def test_arguments_are_checked(fuse):
    chk = fuse.ConsistencyChecker(1.0, 0.01, 10.0, 0.2)
    ref, srcs = S.make_fusion_scene(16, 24, 1, seed=1)
    s = srcs[0]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        chk.check(torch.from_numpy(ref["depth"]), dev(ref["normal"]), ref["K"], ref["E"], dev(s["depth"]),
                  dev(s["normal"]), s["K"], s["E"], dev(ref["confidence"]))
    with pytest.raises(ValueError):
        chk.check(dev(ref["depth"]), dev(ref["normal"][:, :, :2]), ref["K"], ref["E"], dev(s["depth"]),
                  dev(s["normal"]), s["K"], s["E"], dev(ref["confidence"]))
    with pytest.raises(ValueError):
        fuse.ConsistencyChecker(1.0, 0.01, 10.0, 0.2, implement="cupy")


# ----------------------------------------------------------------------------------------------------------------
# Row N2: PFM products written asynchronously (device flip -> pinned D2H -> writer thread) and read back to the device
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("h,w", [(37, 53), (64, 96), (1856, 2752)])
def test_pfm_writer_files_are_byte_identical(tmp_path, h, w):
    from deep3d_aerial_amd import predict as P

    rng = np.random.default_rng(h)
    views = [(rng.standard_normal((h, w)).astype(np.float32), rng.uniform(size=(h, w)).astype(np.float32))
             for _ in range(5)]  # more views than staging slots: slots are reused
    with P.PfmWriter(h, w, 2, depth=2) as wr:
        for i, (d, p) in enumerate(views):
            wr.submit([dev(d), dev(p)[None]], [str(tmp_path / ("a%d_init.pfm" % i)), str(tmp_path / ("a%d_prob.pfm" % i))])
    for i, (d, p) in enumerate(views):
        P.save_pfm(str(tmp_path / "ref_init.pfm"), d)
        P.save_pfm(str(tmp_path / "ref_prob.pfm"), p)
        assert (tmp_path / ("a%d_init.pfm" % i)).read_bytes() == (tmp_path / "ref_init.pfm").read_bytes()
        assert (tmp_path / ("a%d_prob.pfm" % i)).read_bytes() == (tmp_path / "ref_prob.pfm").read_bytes()
        back = P.load_pfm_device(str(tmp_path / ("a%d_init.pfm" % i)))
        assert back.is_cuda and np.array_equal(host(back), d)
        assert np.array_equal(P.load_pfm(str(tmp_path / ("a%d_prob.pfm" % i)))[0], p)


def test_predict_views_products_feed_fusion(tmp_path, fuse):
    """predict.py:126-183 through the async writer: files equal the synchronous writer's, and the kept device maps are
    what the fusion step reads back from disk."""
    from deep3d_aerial_amd import predict as P

    model = P.build_model("casmvsnet", 64)
    S.fill_state_dict_(model.state_dict(), 3)
    model = model.cuda().eval()
    ds = P.SyntheticBlock(3, 3, 64, 96, 64, seed=4)
    kept = P.predict_views(model, ds, str(tmp_path), keep_maps=True)
    assert len(kept) == 3
    for name, (depth, prob) in kept.items():
        on_disk = P.load_pfm_device(str(tmp_path / (name + "_init.pfm")))
        assert torch.equal(on_disk, depth)
        assert np.array_equal(P.load_pfm(str(tmp_path / (name + "_prob.pfm")))[0], host(prob))
        assert (tmp_path / (name + ".txt")).exists()


def test_check_poisoned_inputs_match_oracle(fuse, oracle):
    """Non-finite and absurd depths (NaN, inf, 1e30, negative, denormal) in either map: no fault, the same index
    arithmetic as the oracle (float -> int64 of a non-finite value gives INT64_MIN, indices wrap), identical outputs."""
    ref, srcs = S.make_fusion_scene(64, 80, 1, seed=31)
    s = srcs[0]
    rng = np.random.default_rng(7)
    poison = np.array([np.nan, np.inf, -np.inf, 1e30, -5.0, 1e-40, 0.0, 3e9], np.float32)
    for arr in (ref["depth"], s["depth"]):
        idx = rng.integers(0, arr.size, 400)
        arr.reshape(-1)[idx] = poison[rng.integers(0, len(poison), 400)]
    s["normal"].reshape(-1, 3)[rng.integers(0, s["normal"].shape[0] * s["normal"].shape[1], 50)] = 0.0  # zero normals: 0/0
    with np.errstate(all="ignore"):
        want = oracle.fusion.consistency_check(ref["depth"], ref["normal"], ref["K"], ref["E"], s["depth"], s["normal"],
                                               s["K"], s["E"], ref["confidence"], 1.0, 0.01, 10.0, 0.2)
    got = fuse.ConsistencyChecker(1.0, 0.01, 10.0, 0.2).check(dev(ref["depth"]), dev(ref["normal"]), ref["K"], ref["E"],
                                                               dev(s["depth"]), dev(s["normal"]), s["K"], s["E"],
                                                               dev(ref["confidence"]))
    torch.cuda.synchronize()
    for name, a, b in zip(("mask", "depth_reprojected", "depth_src", "xyz_world_src", "angle"), got, want):
        assert np.array_equal(host(a), b, equal_nan=(name != "mask")), name
    assert want[0].any()


def test_fuse_block_chains_filtered_maps(fuse, oracle):
    """fuse_block = the view loop of fuse_depths on resident maps: every reference view is fused against the maps as
    the EARLIER reference views left them (sources filtered by each check, a reference's own map cut to its final mask).
    Checked against the same chain built from the CPU oracle."""
    ref, srcs = S.make_fusion_scene(72, 96, 3, seed=41)
    cams = [ref] + srcs
    names = ["v%d" % i for i in range(4)]
    views = {n: {"depth": dev(c["depth"]), "normal": dev(c["normal"]), "K": c["K"], "E": c["E"], "id": 10 + i,
                 "confidence": dev(ref["confidence"]) if i == 0 else None}
             for i, (n, c) in enumerate(zip(names, cams))}
    pairs = [{"ref": "v0", "src": ["v1", "v2", "v3", "missing"]}, {"ref": "v1", "src": ["v0", "v2"]}]
    chk = fuse.ConsistencyChecker(1.0, 0.01, 10.0, 0.2)
    got = fuse.fuse_block(views, pairs, chk, fusion_num=10, min_geo_consist_num=2)
    # the same chain on the CPU
    depth = {n: c["depth"].copy() for n, c in zip(names, cams)}
    conf = {"v0": ref["confidence"], "v1": np.ones_like(srcs[0]["depth"])}
    cam = dict(zip(names, cams))
    for pair, g in zip(pairs, got):
        r = cam[pair["ref"]]
        xyz, cs, cnt, nw = oracle.fusion.fusion_ref_init(depth[pair["ref"]], r["normal"], r["K"], r["E"])
        for sn in pair["src"]:
            if sn not in cam:
                continue
            s = cam[sn]
            m, _, dso, pts, ang = oracle.fusion.consistency_check(depth[pair["ref"]], r["normal"], r["K"], r["E"], depth[sn],
                                                                  s["normal"], s["K"], s["E"], conf[pair["ref"]], 1.0, 0.01,
                                                                  10.0, 0.2)
            oracle.fusion.fusion_accumulate(m, pts, ang, 10 + names.index(sn), cnt, xyz, cs)
            depth[sn] = dso
        avg, fm = oracle.fusion.fusion_finalize(xyz, cs, cnt, 2)
        depth[pair["ref"]] = np.where(fm, depth[pair["ref"]], np.float32(0))
        assert np.array_equal(host(g["final_mask"]), fm) and 0.02 < fm.mean() < 0.98
        assert np.array_equal(host(g["avg_xyz_world"]), avg, equal_nan=True)
        assert len(g["vis_infos"]) == 1 + sum(sn in cam for sn in pair["src"])


# ---- row N1 tail: vertices of a reference view (fuse/fusion_3d_normal.py:545-570) ------------------------------------
@pytest.mark.parametrize("h,w,n_vis,skip_line", [(37, 53, 5, 2), (64, 96, 3, 1), (301, 517, 11, 3), (2752, 1856, 11, 2)])
def test_extract_points_bit_exact_vs_oracle(fuse, oracle, h, w, n_vis, skip_line):
    from oracle import fusion as F
    from test_fusion_oracle import _points_scene

    avg, mask, vis, color, normal, sr = _points_scene(h, w, n_vis, 100 + h)
    got = fuse.extract_points(torch.from_numpy(avg).cuda(), torch.from_numpy(mask).cuda(), [torch.from_numpy(v).cuda() for v in vis],
                              torch.from_numpy(color).cuda(), torch.from_numpy(normal).cuda(), sr, skip_line=skip_line)
    xyz, oc, on, ov, onv = F.fusion_points(avg, mask, vis, color, normal, skip_line, sr)
    assert got["n_valid"] == int(mask.sum()) and got["xyz"].shape[0] == xyz.shape[0] > 50
    assert np.array_equal(got["xyz"].cpu().numpy(), xyz) and np.array_equal(got["color"].cpu().numpy(), oc)
    assert np.array_equal(got["normal"].cpu().numpy(), on)
    assert np.array_equal(got["views"].cpu().numpy(), ov) and np.array_equal(got["nviews"].cpu().numpy(), onv)


def test_extract_points_after_view_fusion(fuse):
    """End of the chain: ViewFusion.finalize -> extract_points; few confirmed pixels give no vertices (:541-543)."""
    ref, srcs = S.make_fusion_scene(48, 64, n_src=3, seed=5)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    vf = fuse.ViewFusion(fuse.ConsistencyChecker(1.0, 0.01, 90.0, 0.0), d(ref["depth"]), d(ref["normal"]), ref["K"], ref["E"],
                         d(ref["confidence"]), 1)
    for i, s in enumerate(srcs):
        vf.add_source(d(s["depth"]), d(s["normal"]), s["K"], s["E"], i + 2)
    avg, fm = vf.finalize(2)
    pts = fuse.extract_points(avg, fm, vf.vis_infos, None, vf.normal_world, [-1e9, 1e9, -1e9, 1e9], skip_line=2)
    n_valid = int(fm.sum().item())
    assert pts["n_valid"] == n_valid and pts["xyz"].shape[0] == (n_valid + 1) // 2 and pts["color"] is None
    assert int(pts["nviews"].min().item()) >= 2                       # the reference view + >= 1 confirming source (geo_mask_sum starts at 1)
    assert bool((pts["views"][:, 0] == 0).all())                      # view ids are 0-based in the vertices
    none = fuse.extract_points(avg, torch.zeros_like(fm), vf.vis_infos, None, None, [-1e9, 1e9, -1e9, 1e9])
    assert none["xyz"].shape[0] == 0
