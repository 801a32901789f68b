"""numpy front-end of libfusion_oracle.so (see fusion_oracle.c for the citations and the pinning statement).

TEST INFRASTRUCTURE ONLY -- never imported by deep3d_aerial_amd.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# D3D_FUSION_ORACLE_SO: a pre-built variant (the -fsanitize=address,undefined build of `make -C oracle asan`)
_SO = os.environ.get("D3D_FUSION_ORACLE_SO") or os.path.join(_HERE, "libfusion_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "fusion_oracle.c")
    if os.environ.get("D3D_FUSION_ORACLE_SO"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfusion_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def camera_block(K_ref, E_ref, K_src, E_src):
    """The 94-double camera block: every matrix formed in float32 with numpy.linalg.inv / matmul, as
    consistency_check_n.py:53,58,77,81,102,106 form them with cupy, then widened (exact)."""
    f = lambda a: np.asarray(a, dtype=np.float32)
    K_ref, E_ref, K_src, E_src = f(K_ref), f(E_ref), f(K_src), f(E_src)
    parts = [np.linalg.inv(K_ref), np.matmul(E_src, np.linalg.inv(E_ref))[:3, :4], K_src, np.linalg.inv(K_src),
             np.linalg.inv(E_src), E_ref[:3, :4], K_ref, np.linalg.inv(E_src[:3, :3]), np.linalg.inv(E_ref[:3, :3])]
    assert all(p.dtype == np.float32 for p in parts)
    cam = np.concatenate([p.reshape(-1).astype(np.float64) for p in parts])
    assert cam.size == 94
    return cam


def _ptr(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def consistency_check(depth_ref, normal_ref, K_ref, E_ref, depth_src, normal_src, K_src, E_src, prob_ref,
                      position_threshold, depth_threshold, normal_threshold_deg, confidence_threshold):
    """ConsistencyChecker(position, depth, normal[deg], confidence).check(...) -> (mask, depth_reprojected,
    depth_src with consistent samples zeroed, xyz_world_src [3,H,W], angle_conf [3,H,W])."""
    depth_ref, normal_ref, prob_ref = _f(depth_ref), _f(normal_ref), _f(prob_ref)
    depth_src, normal_src = _f(depth_src), _f(normal_src)
    H, W = depth_ref.shape
    Hs, Ws = depth_src.shape
    cam = camera_block(K_ref, E_ref, K_src, E_src)
    mask = np.zeros((H, W), np.uint8)
    drep = np.zeros((H, W), np.float32)
    dso = np.zeros((Hs, Ws), np.float32)
    xyz = np.zeros((3, H, W), np.float32)
    ang = np.zeros((3, H, W), np.float32)
    fp, dp = ctypes.c_float, ctypes.c_double
    lib().d3d_oracle_consistency_check(
        _ptr(depth_ref, fp), _ptr(normal_ref, fp), _ptr(prob_ref, fp), _ptr(depth_src, fp), _ptr(normal_src, fp),
        _ptr(cam, dp), H, W, Hs, Ws, dp(float(position_threshold)), fp(np.float32(depth_threshold)),
        fp(np.float32(math.cos(math.radians(normal_threshold_deg)))), fp(np.float32(confidence_threshold)),
        _ptr(mask, ctypes.c_ubyte), _ptr(drep, fp), _ptr(dso, fp), _ptr(xyz, fp), _ptr(ang, fp))
    return mask.astype(bool), drep, dso, xyz, ang


def fusion_ref_init(depth_ref, normal_ref, K_ref, E_ref):
    """fusion_3d_normal.py:452-474 -> (all_xyz_world [3,H,W], conf_sum [H,W], geo_mask_sum [H,W], normal_world [H,W,3])."""
    depth_ref, normal_ref = _f(depth_ref), _f(normal_ref)
    H, W = depth_ref.shape
    cam = camera_block(K_ref, E_ref, K_ref, E_ref)  # ESI slot = inv(E_ref)
    xyz = np.zeros((3, H, W), np.float32)
    conf = np.zeros((H, W), np.float32)
    cnt = np.zeros((H, W), np.int32)
    nw = np.zeros((H, W, 3), np.float32)
    fp = ctypes.c_float
    lib().d3d_oracle_fusion_ref_init(_ptr(depth_ref, fp), _ptr(normal_ref, fp), _ptr(cam, ctypes.c_double), H, W,
                                     _ptr(xyz, fp), _ptr(conf, fp), _ptr(cnt, ctypes.c_int32), _ptr(nw, fp))
    return xyz, conf, cnt, nw


def fusion_accumulate(mask, xyz_world_src, angle_conf, src_idx, geo_mask_sum, all_xyz_world, conf_sum):
    """fusion_3d_normal.py:513-518, in place on the three accumulators; returns the visibility plane."""
    H, W = mask.shape
    m8 = np.ascontiguousarray(mask, dtype=np.uint8)
    vis = np.zeros((H, W), np.int32)
    fp = ctypes.c_float
    lib().d3d_oracle_fusion_accumulate(_ptr(m8, ctypes.c_ubyte), _ptr(_f(xyz_world_src), fp), _ptr(_f(angle_conf), fp), H,
                                       W, int(src_idx), _ptr(geo_mask_sum, ctypes.c_int32), _ptr(all_xyz_world, fp),
                                       _ptr(conf_sum, fp), _ptr(vis, ctypes.c_int32))
    return vis


def fusion_finalize(all_xyz_world, conf_sum, geo_mask_sum, min_geo_consist_num):
    """fusion_3d_normal.py:522-527 -> (avg_xyz_world [3,H,W], final_mask [H,W])."""
    H, W = conf_sum.shape
    avg = np.zeros((3, H, W), np.float32)
    fm = np.zeros((H, W), np.uint8)
    fp = ctypes.c_float
    lib().d3d_oracle_fusion_finalize(_ptr(all_xyz_world, fp), _ptr(conf_sum, fp), _ptr(geo_mask_sum, ctypes.c_int32), H, W,
                                     int(min_geo_consist_num), _ptr(avg, fp), _ptr(fm, ctypes.c_ubyte))
    return avg, fm.astype(bool)


def fusion_points(avg_xyz_world, final_mask, vis_infos, color, normal_world, skip_line, scene_range):
    """fusion_3d_normal.py:545-570 -> (xyz [n,3], color [n,3] int32, normal [n,3], views [n,n_vis] int32 (-1 padded), nviews [n])."""
    avg = _f(avg_xyz_world)
    _, H, W = avg.shape
    m8 = np.ascontiguousarray(np.asarray(final_mask).astype(np.uint8))
    vis = [np.ascontiguousarray(v, dtype=np.int32) for v in vis_infos]
    vp = (ctypes.POINTER(ctypes.c_int32) * len(vis))(*[v.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) for v in vis])
    col = None if color is None else _f(color)
    nrm = None if normal_world is None else _f(normal_world)
    sr = (ctypes.c_double * 4)(*[float(x) for x in scene_range[:4]])
    fp, ip = ctypes.c_float, ctypes.c_int32
    f = lib().d3d_oracle_fusion_points
    f.restype = ctypes.c_int64
    null = lambda t: ctypes.cast(None, ctypes.POINTER(t))
    args = [_ptr(avg, fp), _ptr(m8, ctypes.c_ubyte), vp, len(vis), null(fp) if col is None else _ptr(col, fp),
            null(fp) if nrm is None else _ptr(nrm, fp), H, W, int(skip_line), sr]
    n = int(f(*args, null(fp), null(ip), null(fp), null(ip), null(ip)))
    xyz = np.empty((n, 3), np.float32)
    oc = np.empty((n, 3), np.int32)
    on = np.empty((n, 3), np.float32)
    ov = np.empty((n, len(vis)), np.int32)
    onv = np.empty((n,), np.int32)
    if n:
        f(*args, _ptr(xyz, fp), _ptr(oc, ip), _ptr(on, fp), _ptr(ov, ip), _ptr(onv, ip))
    return xyz, (oc if col is not None else None), (on if nrm is not None else None), ov, onv
