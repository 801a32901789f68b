"""Row N3, CPU side: the host camera arithmetic of the item builder against the outputs of the
reference's own preprocess functions (tests/golden/dataset_preprocess.npz, made by tests/golden/make_golden_dataset.py),
the matrix statements of cas_normal_eval.py:134-173 by construction, and the feature cache's bookkeeping."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from conftest import load_golden
from deep3d_aerial_amd import dataset as DS


def test_crop_and_camera_match_reference():
    g = load_golden("dataset_preprocess")
    for i in range(int(g["n_cases"])):
        k = "c%d_" % i
        img, cam = g[k + "img"], g[k + "cam"]
        max_h, max_w = (int(v) for v in g[k + "max_hw"])
        win = DS.crop_window(img.shape[0], img.shape[1], max_h, max_w)
        y0, x0, H, W = DS.slice_window(img.shape[0], img.shape[1], win)
        assert np.array_equal(img[y0:y0 + H, x0:x0 + W], g[k + "cropped"]), i
        ccam = DS.crop_camera(cam.copy(), win[0], win[1])
        assert np.array_equal(ccam, g[k + "crop_cam"]), i
        assert np.array_equal(DS.scale_camera(ccam, 0.5), g[k + "scaled_cam"]), i
    # the fixture holds the small-image case whose slice start is negative
    assert any(g["c%d_cropped" % i].shape[0] < 32 for i in range(int(g["n_cases"])))


def test_stage_projections_by_construction():
    rng = np.random.default_rng(1)
    cams = []
    for _ in range(3):
        cam = np.zeros((2, 4, 4), np.float32)
        cam[0] = np.eye(4)
        cam[0, :3, :] = rng.standard_normal((3, 4))
        cam[1, :3, :3] = [[500, 0, 320], [0, 510, 240], [0, 0, 1]]
        cams.append(cam)
    pm, im = DS.stage_projections(cams)
    for v, cam in enumerate(cams):
        want = cam[0].copy()
        want[:3, :4] = cam[1, :3, :3] @ cam[0, :3, :4]
        assert np.array_equal(pm["stage3"][v], want)
        assert np.array_equal(pm["stage2"][v][:2], want[:2] / 2) and np.array_equal(pm["stage2"][v][2:], want[2:])
        assert np.array_equal(pm["stage1"][v][:2], want[:2] / 4) and np.array_equal(pm["stage1"][v][2:], want[2:])
        assert np.array_equal(im["stage1"][v][:2], cam[1, :2, :3] / 4)
    assert pm["stage1"].dtype == np.float32 and pm["stage1"].shape == (3, 4, 4)


def test_feature_cache_is_lru_and_bounded():
    def pyr(n):
        return {"stage1": torch.zeros(n, dtype=torch.float32)}

    c = DS.FeatureCache(max_bytes=100 * 4)
    c.put("a", pyr(40))
    c.put("b", pyr(40))
    assert c.get("a") is not None  # a is now the most recent
    c.put("c", pyr(40))            # evicts b
    assert "b" not in c and "a" in c and "c" in c and c.bytes == 320
    c.put("huge", pyr(1000))       # larger than the budget: not kept
    assert "huge" not in c and len(c) == 2
    assert c.get("zzz") is None and (c.hits, c.misses) == (1, 1)
    c.clear()
    assert len(c) == 0 and c.bytes == 0


def test_extract_features_uses_the_cache():
    calls = []

    def net(x):
        calls.append(float(x.sum()))
        return {"stage3": x * 2}

    imgs = torch.arange(2 * 3 * 3 * 4 * 4, dtype=torch.float32).reshape(2, 3, 3, 4, 4)
    cache = DS.FeatureCache(1 << 20)
    a = DS.extract_features(net, imgs, ["i0", "i1", "i2"], cache)
    assert len(calls) == 3
    b = DS.extract_features(net, [None, imgs[:, 1], None], ["i0", "i1", "i2"], cache)
    assert len(calls) == 3 and all(torch.equal(x["stage3"], y["stage3"]) for x, y in zip(a, b))
    with pytest.raises(KeyError):
        DS.extract_features(net, [None, None, None], ["i0", "i1", "new"], cache)
    assert len(DS.extract_features(net, imgs)) == 3 and len(calls) == 6  # no keys: every view recomputed


def test_content_matching_reuses_only_identical_images():
    calls = []

    def net(x):
        calls.append(1)
        return {"stage3": x + 1}

    torch.manual_seed(0)
    a, b = torch.randn(1, 3, 16, 24), torch.randn(1, 3, 16, 24)
    c = a.clone()
    c[0, 1, 5, 7] += 1e-6  # one pixel differs: must not be taken for `a`
    cache = DS.FeatureCache(1 << 24, by_content=True)
    f1 = DS.extract_features(net, torch.stack([a, b, a.clone()], dim=1)[0:1].reshape(1, 3, 3, 16, 24), None, cache)
    assert len(calls) == 2 and torch.equal(f1[0]["stage3"], f1[2]["stage3"]) and "_image" not in f1[2]
    f2 = DS.extract_features(net, torch.stack([c, b], dim=1).reshape(1, 2, 3, 16, 24), None, cache)
    assert len(calls) == 3 and torch.equal(f2[0]["stage3"], c + 1) and torch.equal(f2[1]["stage3"], b + 1)
    # without by_content nothing is reused
    plain = DS.FeatureCache(1 << 24)
    DS.extract_features(net, torch.stack([a, a], dim=1).reshape(1, 2, 3, 16, 24), None, plain)
    assert len(calls) == 5


def test_read_image_u8_roundtrip(tmp_path):
    from PIL import Image

    rng = np.random.default_rng(2)
    rgb = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / "a.png")
    got = DS.read_image_u8(str(tmp_path / "a.png"))
    assert got.dtype == np.uint8 and got.flags["C_CONTIGUOUS"] and np.array_equal(got, rgb)
    Image.fromarray(rgb[:, :, 0]).save(tmp_path / "g.png")
    assert DS.read_image_u8(str(tmp_path / "g.png")).shape == (20, 30, 1)


# ----------------------------------------------------------------------------------------
# the block reader against the REFERENCE's dataset class (tests/golden/make_golden_block.py)
# ----------------------------------------------------------------------------------------
def _block(tmp_path):
    import block_fixture as BF

    return BF, BF.write_block(str(tmp_path / "block"))


@pytest.mark.parametrize("normalize", ["mean", "standard"])
def test_block_reader_equals_reference_items(tmp_path, normalize):
    """viewpair.txt / images.txt / cameras.txt / image_path.txt + PNGs -> the item dict of cas_normal_eval.py:94-182:
    cameras, projection pyramids, crops and records EXACTLY as the reference produced them; host-normalised images too
    (same NumPy statements)."""
    from deep3d_aerial_amd import dataset as D

    BF, folder = _block(tmp_path)
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "block_items.npz"))
    ds = D.MVSDataset(folder, "val", BF.VIEW_NUM, normalize, BF.Args())
    assert len(ds) == int(g["n_%s" % normalize]) == 4          # the view without sources is dropped
    assert ds.sample_list[3] == [3, 4, 4, 4]                   # short source lists are padded with their first entry
    for i in range(len(ds)):
        it, k = ds[i], "%s_%d_" % (normalize, i)
        for st in ("stage1", "stage2", "stage3"):
            assert it["proj_matrices"][st].dtype == g[k + "proj_" + st].dtype
            assert np.array_equal(it["proj_matrices"][st], g[k + "proj_" + st])
            assert np.array_equal(it["intri_matrices"][st], g[k + "intri_" + st])
        assert np.array_equal(it["depth_values"], g[k + "depth_values"])
        assert np.array_equal(it["outcam"], g[k + "outcam"])
        assert np.array_equal(it["outimage"], g[k + "outimage"])
        assert list(it["outlocation"]) == list(g[k + "outlocation"])
        assert os.path.basename(it["ref_image_path"]) == str(g[k + "ref_name"])
        assert it["imgs"].dtype == np.float32 and it["imgs"].shape == g[k + "imgs"].shape
        assert np.abs(it["imgs"] - g[k + "imgs"]).max() <= 2e-5


def test_block_reader_device_items_describe_the_same_views(tmp_path):
    """device_item(i): the decoded images + crop windows + cache keys select exactly the pixels of dataset[i]."""
    from deep3d_aerial_amd import dataset as D

    BF, folder = _block(tmp_path)
    ds = D.MVSDataset(folder, "val", BF.VIEW_NUM, "mean", BF.Args())
    for i in range(len(ds)):
        host, dev = ds[i], D.DeviceItems(ds)[i]
        assert len(dev["images_u8"]) == BF.VIEW_NUM and len(set(dev["image_keys"])) == len(set(ds.sample_list[i][:BF.VIEW_NUM]))
        y0, x0, H, W = dev["crop_windows"][0]
        assert np.array_equal(dev["images_u8"][0][y0:y0 + H, x0:x0 + W], host["outimage"])
        for st in ("stage1", "stage2", "stage3"):
            assert np.array_equal(dev["proj_matrices"][st], host["proj_matrices"][st])
        assert np.array_equal(dev["outcam"], host["outcam"]) and dev["outlocation"] == host["outlocation"]


def test_scale_image_unit_scale_is_identity_and_half_scale_averages():
    from deep3d_aerial_amd import dataset as D

    img = np.arange(4 * 6 * 3, dtype=np.uint8).reshape(4, 6, 3)
    assert np.array_equal(D.scale_image(img, 1.0), img)
    half = D.scale_image(img, 0.5)      # INTER_LINEAR at 1/2: each output pixel is the mean of a 2 x 2 block
    want = np.floor(img.reshape(2, 2, 3, 2, 3).astype(np.float64).mean(axis=(1, 3)) + 0.5).astype(np.uint8)
    assert half.shape == (2, 3, 3) and np.array_equal(half, want)


def test_scale_image_follows_the_given_scale_when_the_output_size_is_rounded():
    """cv2.resize(image, None, fx=s, fy=s): the output size is round(n*s) but the source coordinate is
    (dst + 0.5)/s - 0.5 with the GIVEN s (preprocess.py:41-46), not n_in/n_out.  5 columns at s = 0.7 -> 4 columns
    (3.5 rounds to 4); source coordinates 0.2143, 1.6429, 3.0714 and 4.5 (clamped to the last column)."""
    from deep3d_aerial_amd import dataset as D

    row = np.array([[10.0, 20.0, 40.0, 80.0, 160.0]], dtype=np.float32)
    out = D.scale_image(row, 0.7)
    assert out.shape == (1, 4)
    xs = (np.arange(4) + 0.5) / 0.7 - 0.5
    want = np.interp(xs, np.arange(5), row[0])            # np.interp clamps at the ends like the border rule
    assert np.allclose(out[0], want, rtol=0, atol=1e-4)
    assert abs(out[0, 0] - (10 + 0.2142857 * 10)) < 1e-3 and out[0, 3] == 160.0
    ratio_rule = np.interp((np.arange(4) + 0.5) / (4 / 5.0) - 0.5, np.arange(5), row[0])
    assert not np.allclose(out[0], ratio_rule, atol=0.5)   # the n_out/n_in rule gives other values here


def test_feature_cache_hit_survives_eviction_by_an_earlier_miss():
    """An item whose cached view sits BEHIND an uncached one, with room for about one pyramid: the put() for the miss
    evicts the cached key; the pyramid taken in the first pass must still be used (advisor finding, round 1)."""
    calls = []

    def net(x):
        calls.append(float(x.flatten()[0]))
        return {"stage1": x * 2.0}

    cache = DS.FeatureCache(max_bytes=4 * 16 + 8)           # one 16-float pyramid
    a, b = torch.full((1, 1, 4, 4), 1.0), torch.full((1, 1, 4, 4), 2.0)
    DS.extract_features(net, [a], ["A"], cache)             # A cached
    feats = DS.extract_features(net, [b, None], ["B", "A"], cache)   # B uncached and first; A not uploaded (it was cached)
    assert float(feats[0]["stage1"][0, 0, 0, 0]) == 4.0 and float(feats[1]["stage1"][0, 0, 0, 0]) == 2.0
    assert calls == [1.0, 2.0]
    feats = DS.extract_features(net, [b, None, None], ["B", "B", "B"], cache)   # padded source lists repeat an image
    assert len(calls) == 2 and all(float(f["stage1"][0, 0, 0, 0]) == 4.0 for f in feats)


def test_view_records_list_the_fusion_sources(tmp_path):
    """dataset.view_records (pipeline.predict_and_fuse): the view list of the fusion step as fuse/fusion_3d_normal.py:227-249 reads
    it -- every listed source up to fusion_num (not predict's view_num), short lists filled with their first source, views without
    sources dropped, 1-based image positions as visibility ids."""
    from deep3d_aerial_amd import dataset as D, predict as P

    BF, folder = _block(tmp_path)
    ds = D.MVSDataset(folder, "val", BF.VIEW_NUM, "mean", BF.Args())
    recs = D.DeviceItems(ds).view_records(4)
    assert [r["name"] for r in recs] == ["img_00", "img_01", "img_02", "img_03"] and [r["id"] for r in recs] == [1, 2, 3, 4]
    assert recs[0]["src"] == ["img_01", "img_02", "img_03", "img_01"]          # 3 listed, filled to 4
    assert recs[2]["src"] == ["img_01", "img_03", "img_00", "img_04"]          # all 4 listed (predict uses the first view_num - 1)
    assert recs[3]["src"] == ["img_04"] * 4
    assert [r["src"] for r in ds.view_records(2)] == [["img_01", "img_02"], ["img_00", "img_02"], ["img_01", "img_03"], ["img_04", "img_04"]]
    strip = P.SyntheticStrip(5, 3, 32, 32, 64).view_records(10)
    assert strip[4] == {"name": "view_0004", "src": ["view_0000", "view_0001", "view_0002", "view_0003"], "id": 5, "image": 4}


def test_read_scene_blocks(tmp_path):
    """blocks.txt as IO/params_io.py:430-444 writes it and fuse/fusion_3d_normal.py:252-272 reads it."""
    from deep3d_aerial_amd import dataset as D

    p = tmp_path / "blocks.txt"
    p.write_text("2\n-10.5000 20.0000 -3.0000 4.0000 100.0000 200.0000 \n0 1 2 \n0.0000 1.0000 0.0000 1.0000 0.0000 1.0000 \n2 3 \n")
    b = D.read_scene_blocks(str(p))
    assert b == [{"scene_range": [-10.5, 20.0, -3.0, 4.0, 100.0, 200.0], "refs": [0, 1, 2]},
                 {"scene_range": [0.0, 1.0, 0.0, 1.0, 0.0, 1.0], "refs": [2, 3]}]
