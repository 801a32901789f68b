"""BASELINE config 5 end to end: sharded predict -> sharding.all_gather_maps -> fuse.fuse_block ->
fuse.extract_points, two ranks (both on this box's GPU; one per GPU where there are two) against one rank.

Reference: mvs/mvs_cas/predict.py:126-183 (the per-view loop), fuse/fusion_3d_normal.py:404-533 (the fusion loop and its
source-filtering chain), fuse/consistency_check_n.py:141-147."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import pipeline_scene as PS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _launch(n_ranks, out_dir, filter_sources, fuse_partition="views"):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "pipeline_scene.py"), str(out_dir), str(int(filter_sources)), fuse_partition]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-3000:])
    return res.stdout


def _load(folder):
    out = {}
    for root, _, files in sorted(os.walk(folder)):
        for f in sorted(files):
            d = np.load(os.path.join(root, f))
            out[os.path.relpath(os.path.join(root, f[:-4]), folder)] = {k: d[k] for k in d.files}
    return out


def _same(a, b):
    assert sorted(a) == sorted(b)
    for name in a:
        for k in a[name]:
            assert np.array_equal(a[name][k], b[name][k]), (name, k)


def test_two_ranks_fuse_what_one_rank_fuses(tmp_path):
    """filter_sources off: every reference view is fused against the unmodified gathered maps, so the union of the two ranks'
    results -- masks, vertices, normals, visibility lists -- is the single-rank result bit for bit, and so are the PFM products."""
    from deep3d_aerial_amd import predict as P

    out1 = _launch(1, tmp_path / "one", False)
    out2 = _launch(2, tmp_path / "two", False)
    assert "rank 0/2" in out2 and "rank 1/2" in out2 and "rank 0/1" in out1
    one, two = _load(tmp_path / "one" / "fused"), _load(tmp_path / "two" / "fused")
    assert len(one) == PS.N_VIEWS
    _same(one, two)
    # the fixture is not trivial: most reference views confirm thousands of pixels
    confirmed = [int(np.unpackbits(v["final_mask"], axis=1)[:, :PS.W].sum()) for v in one.values()]
    assert sum(c > 1000 for c in confirmed) >= PS.N_VIEWS - 1, confirmed
    assert sum(len(v["xyz"]) for v in one.values()) > 5000
    for f in sorted(os.listdir(tmp_path / "one" / "MVS")):
        assert (tmp_path / "one" / "MVS" / f).read_bytes() == (tmp_path / "two" / "MVS" / f).read_bytes(), f
    d, _ = P.load_pfm(str(tmp_path / "two" / "MVS" / "scene_03_init.pfm"))
    assert np.array_equal(d, PS.SceneViews().views[3]["depth"])


def test_source_filtering_chain_is_kept_inside_a_rank(tmp_path):
    """filter_sources on (the reference's save_temp chain, fusion_3d_normal.py:479-480, 504-510, 529-533): a rank runs the chain
    over ITS reference views in list order from the unfiltered gathered maps -- exactly fuse_block over that rank's pairs -- and
    one rank reproduces the chain over the whole list.  Only the seam between the rank blocks differs from the single-rank run."""
    from deep3d_aerial_amd import fuse, sharding

    _launch(1, tmp_path / "one", True)
    _launch(2, tmp_path / "two", True)
    one, two = _load(tmp_path / "one" / "fused"), _load(tmp_path / "two" / "fused")
    scene = PS.SceneViews()
    recs = scene.view_records(PS.FUSION_NUM)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    views = {r["name"]: {"depth": dev(v["depth"]), "confidence": dev(v["confidence"]), "K": v["K"], "E": v["E"], "id": r["id"]}
             for r, v in zip(recs, scene.views)}
    for world, got in ((1, one), (2, two)):
        for rank in range(world):
            mine = sharding.shard_views(len(recs), rank, world)
            pairs = [{"ref": recs[i]["name"], "src": recs[i]["src"][:PS.FUSION_NUM]} for i in mine]
            for f in fuse.fuse_block(views, pairs, PS.checker(), fusion_num=PS.FUSION_NUM, min_geo_consist_num=3, filter_sources=True):
                fm = np.unpackbits(got[f["ref"]]["final_mask"], axis=1)[:, :PS.W].astype(bool)
                assert np.array_equal(fm, f["final_mask"].cpu().numpy()), (world, rank, f["ref"])
    # rank 0's block starts the chain as the single-rank run does: identical; rank 1's first views see unfiltered sources
    first = sharding.shard_views(len(recs), 0, 2)
    for i in first:
        for k in one[recs[i]["name"]]:
            assert np.array_equal(one[recs[i]["name"]][k], two[recs[i]["name"]][k])
    n1 = sum(len(v["xyz"]) for v in one.values())
    n2 = sum(len(v["xyz"]) for v in two.values())
    assert n1 > 0 and n2 > 0
    print("vertices: one rank %d, two ranks %d" % (n1, n2))
    # and the chain does something on this fixture (otherwise the test above would be this one)
    _launch(1, tmp_path / "off", False)
    noff = sum(len(v["xyz"]) for v in _load(tmp_path / "off" / "fused").values())
    assert noff > n1


def test_scene_blocks_keep_the_reference_chain_for_any_number_of_ranks(tmp_path):
    """fuse_partition="scene_blocks": whole scene blocks of blocks.txt per rank.  The reference starts the source-filtering chain
    afresh for every scene block (fusion_3d_normal.py:586-587, 593-608), so with FILTERING ON two ranks produce the single-rank
    arrays bit for bit -- and those are fuse_block over each block's view list from the unfiltered maps, clipped to the block's range."""
    from deep3d_aerial_amd import fuse

    _launch(1, tmp_path / "one", True, "scene_blocks")
    _launch(2, tmp_path / "two", True, "scene_blocks")
    one, two = _load(tmp_path / "one" / "fused"), _load(tmp_path / "two" / "fused")
    assert sorted(one) == sorted("scene_%d/scene_%02d" % (b, i) for b, blk in enumerate(PS.SCENE_BLOCKS) for i in blk["refs"])
    _same(one, two)
    scene = PS.SceneViews()
    recs = scene.view_records(PS.FUSION_NUM)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    views = {r["name"]: {"depth": dev(v["depth"]), "confidence": dev(v["confidence"]), "K": v["K"], "E": v["E"], "id": r["id"]}
             for r, v in zip(recs, scene.views)}
    for b, blk in enumerate(PS.SCENE_BLOCKS):
        pairs = [{"ref": recs[i]["name"], "src": recs[i]["src"][:PS.FUSION_NUM]} for i in blk["refs"]]
        for f in fuse.fuse_block(views, pairs, PS.checker(), fusion_num=PS.FUSION_NUM, min_geo_consist_num=3, filter_sources=True):
            got = one["scene_%d/%s" % (b, f["ref"])]
            assert np.array_equal(np.unpackbits(got["final_mask"], axis=1)[:, :PS.W].astype(bool), f["final_mask"].cpu().numpy())
            pts = fuse.extract_points(f["avg_xyz_world"], f["final_mask"], f["vis_infos"], None, f["normal_world"], blk["scene_range"], 2)
            assert np.array_equal(got["xyz"], pts["xyz"].cpu().numpy())
    # view 2 is fused in two blocks: unfiltered at the head of block 1, behind views 0 and 1 in block 0; block 1 clips in x
    assert not np.array_equal(one["scene_0/scene_02"]["final_mask"], one["scene_1/scene_02"]["final_mask"])
    x = one["scene_1/scene_03"]["xyz"][:, 0]
    assert len(x) > 100 and x.min() > -60.0 and x.max() < 40.0


def test_predict_main_fuse_flag_two_ranks(tmp_path):
    """`predict.main --fuse` through mvs_dl.MVS_Inference on the block folder, one rank and two: same PFM products, same fused
    arrays (filter off).  The real network with seeded weights: the plumbing of the CLI boundary, not the geometry."""
    import block_fixture as BF
    from deep3d_aerial_amd import mvs_dl, predict as P, synthetic as S

    folder = BF.write_block(str(tmp_path / "block"))
    model = P.build_model("casmvsnet", BF.NUM_DEPTH)
    S.fill_state_dict_(model.state_dict(), 31)
    ckpt = str(tmp_path / "model_000001_0.1000.ckpt")
    torch.save({"epoch": 1, "model": {"module." + k: v for k, v in model.state_dict().items()}, "optimizer": {}}, ckpt)
    kw = dict(view_num=BF.VIEW_NUM, num_depth=BF.NUM_DEPTH, model_type="casmvsnet", pretrain_weight=ckpt,
              extra_args=["--fuse", "--fuse_filter_sources=0", "--geo_consist_num=1", "--depth_threshold=0.5", "--position_threshold=50"])
    one, two = tmp_path / "one" / "MVS", tmp_path / "two" / "MVS"
    mvs_dl.MVS_Inference(BF.MAX_W, BF.MAX_H, **kw).run(folder, str(one))
    mvs_dl.MVS_Inference(BF.MAX_W, BF.MAX_H, n_gpus=2, **kw).run(folder, str(two))
    f1, f2 = _load(one / "fused"), _load(two / "fused")
    assert sorted(f1) == ["img_%02d" % i for i in range(4)]
    _same(f1, f2)
    for f in sorted(p.name for p in one.iterdir() if p.is_file()):
        assert (one / f).read_bytes() == (two / f).read_bytes(), f
