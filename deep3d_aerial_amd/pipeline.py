"""BASELINE config 5 end to end: sharded depth inference -> all-gather of the (depth, confidence) maps -> fusion.

The reference runs the two steps as separate programs that meet on disk: predict.py writes {name}_init.pfm / _prob.pfm /
.txt per reference view (predict.py:146-183), fuse/fusion_3d_normal.py:404-533 reads them back view by view.  Here the
maps never leave HBM:

    rank r:  predict_views(keep_maps=True) over ITS views      (sharding.shard_views; the PFM products are still written)
             all_gather_maps([n_local, 2, H, W])               (the ONE collective of the path: RCCL over xGMI, gloo when
                                                                ranks share a card) + the views' [2,4,4] cameras
             fuse.fuse_block over ITS reference views           (sources looked up in the gathered maps)
             fuse.extract_points per reference view

Ownership rule: the rank that sweeps a reference view also fuses it (the same contiguous blocks of `viewpair.txt` order,
sharding.shard_views), against the gathered maps of ALL views.  What this preserves of the reference:

* With `filter_sources=False` every reference view is fused against the unmodified depth maps: results do not depend on the
  order of the views, so the union over ranks is bit for bit the single-rank result (tests/test_pipeline_gpu.py).
* With `filter_sources=True` (the reference's default, save_temp: fusion_3d_normal.py:417-418, 479-480, 504-510, 529-533)
  a source map loses the samples a reference view has confirmed before the NEXT reference view reads it -- a chain through
  the whole view list of a scene block.  A rank runs that chain over its own reference views in list order, starting from
  the unfiltered gathered maps: inside a rank's block of views the reference's behaviour is kept exactly; samples confirmed
  by reference views of LOWER ranks are still offered (duplicate points along the seams between rank blocks).  One rank
  reproduces the reference's chain over the whole list.

* `fuse_partition="scene_blocks"` (with the scene blocks of blocks.txt, dataset.read_scene_blocks): the reference fuses scene
  block by scene block and starts the chain afresh for each (fusion_3d_normal.py:593-608; the tmp folder is emptied at the end
  of fuse_depths, :586-587).  Scene blocks are therefore the units whose chains are INDEPENDENT: dealt whole to the ranks
  (block b to rank b mod N -- fusion ownership then differs from sweep ownership, which costs nothing: every rank holds every
  map after the gather) they reproduce the reference's result exactly for any N, filtering on, vertices clipped to each block's
  scene range as :557 does.  Balanced only when there are at least N blocks; the default partition ("views") always is.

The view list, the sources of a view (all the sources viewpair.txt lists, up to `fusion_num` -- the fusion step does not stop
at predict's view_num: fusion_3d_normal.py:98, 476) and the 1-based image ids of the visibility lists come from the dataset
(`view_records`); cameras are the `outcam` of the item whose reference view the image is -- what the reference reads back from
{name}.txt (fusion_3d_normal.py:425-427; write_red_cam's str(float32) round-trips exactly).
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import fuse, predict, sharding


def view_records(dataset, fusion_num=10):
    """[{"name", "src": [names], "id"}] for every item of the dataset, by item index -- metadata only (no image is read).
    Datasets provide `view_records(fusion_num)`: dataset.MVSDataset / DeviceItems from viewpair.txt + images.txt,
    predict.SyntheticStrip from its ring."""
    fn = getattr(dataset, "view_records", None)
    if fn is None:
        raise TypeError("%s has no view_records(): the fusion step needs the view list of the block" % type(dataset).__name__)
    recs = fn(fusion_num)
    if len(recs) != len(dataset):
        raise ValueError("view_records() lists %d views, the dataset has %d items" % (len(recs), len(dataset)))
    return recs


def _gather_objects(obj, world_size):
    if world_size == 1:
        return [obj]
    out = [None] * world_size
    dist.all_gather_object(out, obj)
    return out


def predict_and_fuse(model, dataset, output_folder, rank=0, world_size=1, checker=None, fusion_num=10, min_geo_consist_num=4,
                     filter_sources=True, partition="block", scene_range=None, skip_line=2, feature_cache_bytes=0,
                     device="cuda", timings=None, display=False, fuse_partition="views", scene_blocks=None):
    """Runs the three steps above for this rank.  Returns a list, one entry per reference view this rank owns, of
    {"ref", "final_mask" [H,W] bool, "avg_xyz_world" [3,H,W], "points": fuse.extract_points(...) dict} (device tensors).
    timings: dict that receives predict_s, allgather_ms (the collective alone, synchronised on both sides), fuse_s."""
    if checker is None:
        checker = fuse.ConsistencyChecker(1.0, 0.01, 90.0, 0.2)   # Fuse_Depth_Map's defaults (fusion_3d_normal.py:56-57)
    n = len(dataset)
    recs = view_records(dataset, fusion_num)
    mine = sharding.shard_views(n, rank, world_size, partition)
    cams = {}
    t0 = time.perf_counter()
    maps = predict.predict_views(model, dataset, output_folder, rank, world_size, device=device, keep_maps=True,
                                 feature_cache_bytes=feature_cache_bytes, display=display, partition=partition, cams=cams)
    names = [recs[i]["name"] for i in mine]
    if list(maps.keys()) != names:
        raise RuntimeError("the views predict_views produced %s are not this rank's %s" % (list(maps.keys()), names))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    # ---- every rank learns the map size (a rank may own no view) and the views agree on it
    shapes = _gather_objects([tuple(maps[k][0].shape) for k in names], world_size)
    sizes = {s for per in shapes for s in per}
    if len(sizes) != 1:
        raise ValueError("the maps of a block must share one size to be gathered (got %s)" % sorted(sizes))
    H, W = sizes.pop()
    dev = torch.device(device)
    if names:
        local = torch.stack([torch.stack([maps[k][0], maps[k][1]]) for k in names])
        local_cams = torch.from_numpy(np.stack([np.asarray(cams[k], np.float32) for k in names])).to(dev)
    else:
        local = torch.empty((0, 2, H, W), dtype=torch.float32, device=dev)
        local_cams = torch.empty((0, 2, 4, 4), dtype=torch.float32, device=dev)
    if world_size > 1:
        dist.barrier()
    torch.cuda.synchronize()
    g0 = time.perf_counter()
    all_maps = sharding.all_gather_maps(local, n, rank, world_size, partition)       # [n, 2, H, W], by global view index
    torch.cuda.synchronize()
    g1 = time.perf_counter()
    all_cams = sharding.all_gather_maps(local_cams, n, rank, world_size, partition).cpu().numpy()   # [n, 2, 4, 4]
    del local
    # ---- fusion of this rank's reference views against the gathered maps
    views = {}
    for i, r in enumerate(recs):
        views[r["name"]] = {"depth": all_maps[i, 0], "confidence": all_maps[i, 1], "K": all_cams[i, 1, :3, :3].copy(),
                            "E": all_cams[i, 0].copy(), "id": int(r["id"])}
    if fuse_partition not in ("views", "scene_blocks"):
        raise ValueError("fuse_partition must be 'views' or 'scene_blocks'")
    pair_of = lambda i: {"ref": recs[i]["name"], "src": list(recs[i]["src"])[:fusion_num]}
    out = []
    if fuse_partition == "scene_blocks":
        if not scene_blocks:
            raise ValueError("fuse_partition='scene_blocks' needs the scene blocks (dataset.read_scene_blocks(blocks.txt))")
        by_image = {int(r.get("image", i)): i for i, r in enumerate(recs)}
        for b, blk in enumerate(scene_blocks):
            if b % world_size != rank:
                continue
            # (a reference view listed in a block but absent from the view list -- it had no sources -- has no maps: skipped with the
            #  warning the reference gives for a missing PFM, fusion_3d_normal.py:420-422)
            pairs = [pair_of(by_image[i]) for i in blk["refs"] if i in by_image]
            fused = fuse.fuse_block(views, pairs, checker, fusion_num=fusion_num, min_geo_consist_num=min_geo_consist_num,
                                    filter_sources=filter_sources)
            for f in fused:
                pts = fuse.extract_points(f["avg_xyz_world"], f["final_mask"], f["vis_infos"], None, f["normal_world"],
                                          blk["scene_range"], skip_line)
                out.append({"ref": f["ref"], "scene": b, "final_mask": f["final_mask"], "avg_xyz_world": f["avg_xyz_world"], "points": pts})
    else:
        fused = fuse.fuse_block(views, [pair_of(i) for i in mine], checker, fusion_num=fusion_num,
                                min_geo_consist_num=min_geo_consist_num, filter_sources=filter_sources)
        sr = scene_range if scene_range is not None else [-np.inf, np.inf, -np.inf, np.inf]
        for f in fused:
            pts = fuse.extract_points(f["avg_xyz_world"], f["final_mask"], f["vis_infos"], None, f["normal_world"], sr, skip_line)
            out.append({"ref": f["ref"], "final_mask": f["final_mask"], "avg_xyz_world": f["avg_xyz_world"], "points": pts})
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if timings is not None:
        timings.update(views=len(mine), predict_s=t1 - t0, allgather_ms=(g1 - g0) * 1e3,
                       allgather_bytes=int(all_maps.numel() * 4), fuse_s=t2 - g1, map_size=(H, W),
                       backend=dist.get_backend() if world_size > 1 else "none")
    return out


def save_fused(results, folder):
    """One <folder>/<ref>.npz per reference view: final_mask (bit-packed rows), xyz [n,3], normal [n,3], views [n,n_vis] (sorted
    0-based image indices, -1 padded), nviews [n].  The OpenMVS .mvs / .ply writers (IO/mvs_io.py) are out of scope:
    these arrays are what Interface_Fused would be handed (fusion_3d_normal.py:558-570)."""
    os.makedirs(folder, exist_ok=True)
    paths = []
    for r in results:
        p = r["points"]
        sub = os.path.join(folder, "scene_%d" % r["scene"]) if "scene" in r else folder   # (a view can belong to several scene blocks)
        os.makedirs(sub, exist_ok=True)
        path = os.path.join(sub, r["ref"] + ".npz")
        fm = r["final_mask"].cpu().numpy()
        np.savez(path, mask_shape=np.array(fm.shape), final_mask=np.packbits(fm, axis=1), xyz=p["xyz"].cpu().numpy(),
                 normal=p["normal"].cpu().numpy() if p["normal"] is not None else np.zeros((0, 3), np.float32),
                 views=p["views"].cpu().numpy(), nviews=p["nviews"].cpu().numpy(), n_valid=np.array(p["n_valid"]))
        paths.append(path)
    return paths
