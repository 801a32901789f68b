"""Geometric consistency check and per-view fusion accumulators on the GPU.

Host-side mirror of the reference interface for this step:

* `ConsistencyChecker(position_threshold, depth_threshold, normal_threshold, confidence_threshold)` with
  `.check(depth_ref, normal_ref, intrinsics_ref, extrinsics_ref, depth_src, normal_src, intrinsics_src,
  extrinsics_src, prob_map_ref)` -> `(mask, depth_reprojected, depth_src, xyz_world_src, angle_confidence)` --
  fuse/consistency_check_n.py:17-27, 141-147.  The reference's CuPy path uploads nine arrays and downloads five
  per (ref, src) pair; here the maps are device tensors (the depth / confidence maps the plane-sweep path has just
  produced stay resident) and a pair is one kernel launch (`d3d_consistency_check`).
* `ViewFusion` is the body of `Fuse_Depth_Map.fuse_depths` for one reference view (fuse/fusion_3d_normal.py:
  452-474 reference init, :476-518 per-source accumulation, :522-527 average / final mask) on resident
  accumulators; `add_source` is the check fused with the accumulation (`d3d_fusion_accumulate`), so the pair
  outputs never reach memory.

The 3x3 / 4x4 camera algebra (inverses, E_src @ inv(E_ref)) happens on the host in float32 with NumPy, exactly where
and how the reference forms those matrices (np.fromstring(dtype=float32) cameras, linalg.inv keeps float32).
All per-pixel arithmetic runs in libdeep3d_planesweep.so; there is no CPU path.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from .ops import _chk, _stream

_CAM_DOUBLES = 94  # D3D_FUSION_CAM_DOUBLES


def camera_block(intrinsics_ref, extrinsics_ref, intrinsics_src, extrinsics_src):
    """The `cam` argument of d3d_consistency_check (include/deep3d_planesweep.h): nine float32 matrices, widened."""
    def f32(a):
        a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
        return np.ascontiguousarray(a, dtype=np.float32)

    Kr, Er, Ks, Es = f32(intrinsics_ref), f32(extrinsics_ref), f32(intrinsics_src), f32(extrinsics_src)
    if Kr.shape != (3, 3) or Ks.shape != (3, 3) or Er.shape != (4, 4) or Es.shape != (4, 4):
        raise ValueError("intrinsics must be [3,3] and extrinsics [4,4]")
    inv = np.linalg.inv
    parts = [inv(Kr), np.matmul(Es, inv(Er))[:3, :4], Ks, inv(Ks), inv(Es), Er[:3, :4], Kr, inv(Es[:3, :3]),
             inv(Er[:3, :3])]
    cam = np.concatenate([p.reshape(-1).astype(np.float64) for p in parts])
    assert cam.size == _CAM_DOUBLES
    return (ctypes.c_double * _CAM_DOUBLES)(*cam.tolist())


def _map(t, name, shape=None, last=None):
    p = _chk(t, name)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError("%s must be %s (got %s)" % (name, tuple(shape), tuple(t.shape)))
    if last is not None and (t.dim() != 3 or t.shape[2] != last):
        raise ValueError("%s must be [H,W,%d] (got %s)" % (name, last, tuple(t.shape)))
    return p


class ConsistencyChecker(object):
    """fuse/consistency_check_n.py:17-27: thresholds in pixels, relative depth, degrees, confidence."""

    def __init__(self, position_threshold, depth_threshold, normal_threshold, confidence_threshold, implement="hip"):
        self.position_threshold = float(position_threshold)
        self.depth_threshold = float(depth_threshold)
        self.normal_threshold = math.cos(math.radians(normal_threshold))
        self.confidence_threshold = float(confidence_threshold)
        self.implement = implement.lower()
        if self.implement != "hip":
            raise ValueError("this build has one implementation, 'hip' (got %r)" % implement)

    def _thresholds(self):
        return (ctypes.c_double(self.position_threshold), ctypes.c_float(self.depth_threshold),
                ctypes.c_float(self.normal_threshold), ctypes.c_float(self.confidence_threshold))

    def check(self, depth_ref, normal_ref, intrinsics_ref, extrinsics_ref, depth_src, normal_src, intrinsics_src,
              extrinsics_src, prob_map_ref):
        """Device tensors: depth_ref, prob_map_ref [H,W]; normal_ref [H,W,3]; depth_src [Hs,Ws]; normal_src [Hs,Ws,3].
        Returns (mask bool [H,W], depth_reprojected [H,W], depth_src with the consistent samples zeroed [Hs,Ws],
        xyz_world_src [3,H,W], angle_confidence [3,H,W])."""
        if depth_ref.dim() != 2 or depth_src.dim() != 2:
            raise ValueError("depth maps must be [H,W]")
        H, W = depth_ref.shape
        Hs, Ws = depth_src.shape
        dev = depth_ref.device
        cam = camera_block(intrinsics_ref, extrinsics_ref, intrinsics_src, extrinsics_src)
        args = [_map(depth_ref, "depth_ref"), _map(normal_ref, "normal_ref", (H, W, 3)),
                _map(prob_map_ref, "prob_map_ref", (H, W)), _map(depth_src, "depth_src"),
                _map(normal_src, "normal_src", (Hs, Ws, 3))]
        mask = torch.empty((H, W), dtype=torch.uint8, device=dev)
        depth_reprojected = torch.empty((H, W), dtype=torch.float32, device=dev)
        depth_src_out = depth_src.clone()
        xyz_world_src = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        angle = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        rc = _lib.load().d3d_consistency_check(
            *args, cam, H, W, Hs, Ws, *self._thresholds(), ctypes.c_void_p(mask.data_ptr()),
            _chk(depth_reprojected, "depth_reprojected"), _chk(depth_src_out, "depth_src_out"),
            _chk(xyz_world_src, "xyz_world_src"), _chk(angle, "angle_confidence"), _stream())
        _lib.check(rc, "d3d_consistency_check")
        return mask.bool(), depth_reprojected, depth_src_out, xyz_world_src, angle


class ViewFusion(object):
    """Accumulators of one reference view (fuse/fusion_3d_normal.py:446-527).

        vf = ViewFusion(checker, depth_ref, normal_ref, K_ref, E_ref, confidence, ref_idx)
        for each source view:  src_depth_filtered = vf.add_source(depth_src, normal_src, K_src, E_src, src_idx)
        avg_xyz_world, final_mask = vf.finalize(min_geo_consist_num)

    State (device): all_xyz_world [3,H,W], xyz_confidence [H,W] (the reference's three identical planes, kept once),
    geo_mask_sum [H,W] int32, vis_infos: list of [H,W] int32 planes (index 0 = the reference view's own id),
    normal_world [H,W,3] unit world normals of the reference view."""

    def __init__(self, checker, depth_ref, normal_ref, intrinsics_ref, extrinsics_ref, confidence, ref_idx):
        if depth_ref.dim() != 2:
            raise ValueError("depth_ref must be [H,W]")
        self.checker = checker
        self.H, self.W = depth_ref.shape
        H, W = self.H, self.W
        dev = depth_ref.device
        self.depth_ref, self.normal_ref, self.confidence = depth_ref, normal_ref, confidence
        self.K_ref, self.E_ref = intrinsics_ref, extrinsics_ref
        _map(confidence, "confidence", (H, W))
        self.all_xyz_world = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        self.xyz_confidence = torch.empty((H, W), dtype=torch.float32, device=dev)
        self.geo_mask_sum = torch.empty((H, W), dtype=torch.int32, device=dev)
        self.normal_world = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
        cam = camera_block(intrinsics_ref, extrinsics_ref, intrinsics_ref, extrinsics_ref)  # inv(E_src) slot = inv(E_ref)
        rc = _lib.load().d3d_fusion_ref_init(
            _map(depth_ref, "depth_ref"), _map(normal_ref, "normal_ref", (H, W, 3)), cam, H, W,
            _chk(self.all_xyz_world, "all_xyz_world"), _chk(self.xyz_confidence, "xyz_confidence"),
            ctypes.c_void_p(self.geo_mask_sum.data_ptr()), _chk(self.normal_world, "normal_world"), _stream())
        _lib.check(rc, "d3d_fusion_ref_init")
        self.vis_infos = [torch.full((H, W), int(ref_idx), dtype=torch.int32, device=dev)]  # :473

    def add_source(self, depth_src, normal_src, intrinsics_src, extrinsics_src, src_idx, filter_source=True):
        """:476-518 for one source view.  Returns the source depth map with the samples this reference view
        confirmed set to 0 (what the reference writes back as the source's `_init.pfm` when save_temp is on), or
        None with filter_source=False."""
        if depth_src.dim() != 2:
            raise ValueError("depth_src must be [Hs,Ws]")
        H, W = self.H, self.W
        Hs, Ws = depth_src.shape
        dev = self.depth_ref.device
        cam = camera_block(self.K_ref, self.E_ref, intrinsics_src, extrinsics_src)
        vis = torch.empty((H, W), dtype=torch.int32, device=dev)
        out = depth_src.clone() if filter_source else None
        rc = _lib.load().d3d_fusion_accumulate(
            _map(self.depth_ref, "depth_ref"), _map(self.normal_ref, "normal_ref"), _map(self.confidence, "confidence"),
            _map(depth_src, "depth_src"), _map(normal_src, "normal_src", (Hs, Ws, 3)), cam, H, W, Hs, Ws,
            *self.checker._thresholds(), int(src_idx), ctypes.c_void_p(self.geo_mask_sum.data_ptr()),
            _chk(self.all_xyz_world, "all_xyz_world"), _chk(self.xyz_confidence, "xyz_confidence"),
            ctypes.c_void_p(vis.data_ptr()), None if out is None else _chk(out, "depth_src_out"), _stream())
        _lib.check(rc, "d3d_fusion_accumulate")
        self.vis_infos.append(vis)
        return out

    def finalize(self, min_geo_consist_num):
        """:522-527 -> (avg_xyz_world [3,H,W] float32, final_mask [H,W] bool)."""
        H, W = self.H, self.W
        dev = self.depth_ref.device
        avg = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        fm = torch.empty((H, W), dtype=torch.uint8, device=dev)
        rc = _lib.load().d3d_fusion_finalize(
            _chk(self.all_xyz_world, "all_xyz_world"), _chk(self.xyz_confidence, "xyz_confidence"),
            ctypes.c_void_p(self.geo_mask_sum.data_ptr()), H, W, int(min_geo_consist_num), _chk(avg, "avg_xyz_world"),
            ctypes.c_void_p(fm.data_ptr()), _stream())
        _lib.check(rc, "d3d_fusion_finalize")
        return avg, fm.bool()


def extract_points(avg_xyz_world, final_mask, vis_infos, ref_img, normal_world, scene_range, skip_line=2):
    """fuse/fusion_3d_normal.py:545-570 on the device: the vertices of one reference view.

    avg_xyz_world [3,H,W] and final_mask [H,W] from ViewFusion.finalize; vis_infos: ViewFusion.vis_infos (list of [H,W]
    int32 planes holding 1-based image indices, 0 = not visible); ref_img [H,W,3] float32 in 0..1 (read_img, :174-186) or
    None; normal_world [H,W,3] or None; scene_range = [min_x, max_x, min_y, max_y, ...] of the block (:598).
    Returns a dict of device tensors: "xyz" [n,3], "color" [n,3] int32 (or None), "normal" [n,3] (or None), "views"
    [n,n_vis] int32 = the sorted 0-based view indices of every vertex, -1 padded, "nviews" [n] -- the content of the
    reference's total_vertices / total_verticesColor / total_verticesNormal for this view, in the same order.  A view with
    fewer than 10 confirmed pixels yields no vertices (:541-543)."""
    if avg_xyz_world.dim() != 3 or avg_xyz_world.shape[0] != 3:
        raise ValueError("avg_xyz_world must be [3,H,W]")
    _, H, W = avg_xyz_world.shape
    dev = avg_xyz_world.device
    lib = _lib.load()
    fm = final_mask.to(torch.uint8).contiguous()
    if tuple(fm.shape) != (H, W):
        raise ValueError("final_mask must be [H,W]")
    n_vis = len(vis_infos)
    if not 1 <= n_vis <= 64:
        raise ValueError("1..64 visibility planes (got %d)" % n_vis)
    for v in vis_infos:
        if v.dtype != torch.int32 or tuple(v.shape) != (H, W) or not v.is_cuda or not v.is_contiguous():
            raise TypeError("vis_infos must be contiguous CUDA int32 [H,W] tensors")
    scratch = torch.empty((int(lib.d3d_fusion_points_scratch_bytes(H, W)),), dtype=torch.uint8, device=dev)
    keep = torch.empty((H, W), dtype=torch.uint8, device=dev)
    counts = torch.zeros((2,), dtype=torch.int32, device=dev)
    sr = (ctypes.c_double * 4)(*[float(x) for x in list(scene_range)[:4]])
    rc = lib.d3d_fusion_mark_points(_chk(avg_xyz_world, "avg_xyz_world"), ctypes.c_void_p(fm.data_ptr()), H, W, int(skip_line), sr,
                                    ctypes.c_void_p(scratch.data_ptr()), ctypes.c_void_p(keep.data_ptr()),
                                    ctypes.c_void_p(counts.data_ptr()), _stream())
    _lib.check(rc, "d3d_fusion_mark_points")
    n_valid, n = (int(x) for x in counts.tolist())   # the one host read: the caller sizes the outputs
    if n_valid < 10 or n_valid <= 1:                 # :541-543 ("no points left"), :555
        n = 0
    out = {"xyz": torch.empty((n, 3), dtype=torch.float32, device=dev),
           "color": None if ref_img is None else torch.empty((n, 3), dtype=torch.int32, device=dev),
           "normal": None if normal_world is None else torch.empty((n, 3), dtype=torch.float32, device=dev),
           "views": torch.empty((n, n_vis), dtype=torch.int32, device=dev),
           "nviews": torch.empty((n,), dtype=torch.int32, device=dev), "n_valid": n_valid}
    if n == 0:
        return out
    vp = (ctypes.c_void_p * n_vis)(*[v.data_ptr() for v in vis_infos])
    opt = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    rc = lib.d3d_fusion_gather_points(_chk(avg_xyz_world, "avg_xyz_world"), ctypes.c_void_p(keep.data_ptr()), vp, n_vis,
                                      None if ref_img is None else _map(ref_img, "ref_img", (H, W, 3)),
                                      None if normal_world is None else _map(normal_world, "normal_world", (H, W, 3)), H, W,
                                      ctypes.c_void_p(scratch.data_ptr()), _chk(out["xyz"], "xyz"), opt(out["color"]),
                                      opt(out["normal"]), ctypes.c_void_p(out["views"].data_ptr()),
                                      ctypes.c_void_p(out["nviews"].data_ptr()), _stream())
    _lib.check(rc, "d3d_fusion_gather_points")
    return out


def default_normals(h, w, device="cuda"):
    """fuse/fusion_3d_normal.py:441-443, 497-498: views without a normal map get (0, 0, -1) everywhere."""
    n = torch.zeros((h, w, 3), dtype=torch.float32, device=device)
    n[:, :, 2] = -1.0
    return n


def fuse_block(views, pairs, checker, fusion_num=10, min_geo_consist_num=4, filter_sources=True):
    """The view loop of Fuse_Depth_Map.fuse_depths (fuse/fusion_3d_normal.py:404-533) over maps that are already on
    the device (predict.predict_views(keep_maps=True), or sharding.all_gather_maps across ranks) -- no PFM round trip.

    views: dict name -> {"depth" [H,W], "confidence" [H,W] (optional: ones), "normal" [H,W,3] (optional: default),
           "K" [3,3], "E" [4,4] (numpy float32, Tcw), "id" (int written into the visibility planes)}
    pairs: list of {"ref": name, "src": [names...]} (viewpair.txt order).
    With filter_sources=True a source's depth map is replaced by its filtered copy after every check, as the reference
    does through its tmp/ folder when save_temp is on (:417-418, 479-480, 504-510): samples a reference view has confirmed
    are not offered to later reference views again.
    Returns a list of {"ref", "avg_xyz_world" [3,H,W], "final_mask" [H,W] bool, "vis_infos" list of [H,W] int32,
    "normal_world" [H,W,3]} in pair order."""
    depth = {k: v["depth"] for k, v in views.items()}
    out = []
    for pair in pairs:
        r = views[pair["ref"]]
        H, W = depth[pair["ref"]].shape
        dev = depth[pair["ref"]].device
        conf = r.get("confidence")
        if conf is None:
            conf = torch.ones((H, W), dtype=torch.float32, device=dev)
        normal = r.get("normal")
        if normal is None:
            normal = default_normals(H, W, dev)
        vf = ViewFusion(checker, depth[pair["ref"]], normal, r["K"], r["E"], conf, r["id"])
        for name in pair["src"][:fusion_num]:
            if name not in views:
                continue  # the reference warns and skips missing maps (:482-485)
            s = views[name]
            sn = s.get("normal")
            if sn is None:
                sn = default_normals(depth[name].shape[0], depth[name].shape[1], dev)
            filtered = vf.add_source(depth[name], sn, s["K"], s["E"], s["id"], filter_source=filter_sources)
            if filter_sources:
                depth[name] = filtered
        avg, fm = vf.finalize(min_geo_consist_num)
        if filter_sources:  # :529-533: the reference view's own map keeps only its confirmed points
            depth[pair["ref"]] = torch.where(fm, depth[pair["ref"]], torch.zeros_like(depth[pair["ref"]]))
        out.append({"ref": pair["ref"], "avg_xyz_world": avg, "final_mask": fm, "vis_infos": vf.vis_infos,
                    "normal_world": vf.normal_world})
    return out
