"""Input side of a reference view on the GPU.

The reference builds every dataset item on the host (mvs/mvs_cas/datasets/cas_normal_eval.py:94-182): each of the V
images of an item is decoded, cropped, converted to float32 and normalised with NumPy (preprocess.py:60-117), although
neighbouring reference views share most of their images, and then every image goes through the feature network again
(adamvs.py:571-574, cas_mvsnet.py:189-192).  Here

* `crop_window` / `crop_camera` / `stage_projections` are the host-side camera arithmetic of that item builder
  (a few 4x4 matrices: NumPy, as in the reference);
* `center_image` crops and normalises a decoded 8-bit image on the device (`d3d_center_image_u8`: exact integer
  statistics, one upload of h*w*3 bytes instead of h*w*12);
* `FeatureCache` keeps the feature pyramids of recently used images resident in HBM (a 2752x1856 image's three-stage
  pyramid is 286 MB; 288 GB hold hundreds), so an image shared by several reference views is normalised and
  featurised once per block.  The inference drivers take `image_keys=` and consult the cache attached to them.
"""
import collections
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib
from .ops import _chk, _stream


# ----------------------------------------------------------------------------------------
# camera arithmetic of the item builder (host, NumPy)
# ----------------------------------------------------------------------------------------
def crop_window(h, w, max_h, max_w, resize_scale=1, base_image_size=32):
    """preprocess.py:60-77: (start_h, start_w, new_h, new_w) of the centre crop that fits the network (see
    slice_window for the pixels it selects)."""
    max_h = int(max_h * resize_scale)
    max_w = int(max_w * resize_scale)
    new_h = max_h if h > max_h else int(math.ceil(h / base_image_size) * base_image_size)
    new_w = max_w if w > max_w else int(math.ceil(w / base_image_size) * base_image_size)
    start_h = int(math.ceil((h - new_h) / 2))
    start_w = int(math.ceil((w - new_w) / 2))
    return start_h, start_w, new_h, new_w


def slice_window(h, w, window):
    """The pixels `image[start_h:start_h+new_h, start_w:start_w+new_w]` (preprocess.py:78-80) really selects, as
    (y0, x0, H, W): for an image smaller than its rounded-up size the start is negative and Python's slice counts it
    from the end -- kept, because the camera shift of crop_camera uses the same start."""
    start_h, start_w, new_h, new_w = window
    ys = range(h)[slice(start_h, start_h + new_h)]
    xs = range(w)[slice(start_w, start_w + new_w)]
    return (ys.start if len(ys) else 0), (xs.start if len(xs) else 0), len(ys), len(xs)


def crop_camera(cam, start_h, start_w):
    """preprocess.py:81-82: the principal point moves with the crop origin.  cam [2,4,4] (extrinsic; intrinsic +
    depth row), modified in place like the reference and returned."""
    cam[1][0][2] = cam[1][0][2] - start_w
    cam[1][1][2] = cam[1][1][2] - start_h
    return cam


def scale_camera(cam, scale=1):
    """preprocess.py:19-30: focal lengths and principal point times `scale`."""
    new_cam = np.copy(cam)
    rows, cols = [0, 1, 0, 1], [0, 1, 2, 2]  # fx, fy, x0, y0 of the intrinsic block
    new_cam[1, rows, cols] = cam[1, rows, cols] * scale
    return new_cam


def stage_projections(cams, sample_scale=1):
    """cas_normal_eval.py:134-173: per view proj = E with its top 3x4 replaced by K @ E[:3,:4]; the stage-2 / stage-1
    matrices have the first two rows divided by 2 / 4.  cams: list of [2,4,4].  Returns (proj_matrices_ms,
    intri_matrices_ms) with keys stage1..stage3 as the reference's sample dict."""
    projs, intris = [], []
    for cam in cams:
        sc = scale_camera(cam, scale=sample_scale)
        extrinsics = sc[0, :, :]
        intrinsics = sc[1, 0:3, 0:3]
        proj = extrinsics.copy()
        proj[:3, :4] = np.matmul(intrinsics, proj[:3, :4])
        projs.append(proj)
        intris.append(intrinsics)
    projs, intris = np.stack(projs), np.stack(intris)

    def pyramid(m):
        half, quarter = m.copy(), m.copy()
        half[:, :2, :] = m[:, :2, :] / 2
        quarter[:, :2, :] = m[:, :2, :] / 4
        return {"stage1": quarter, "stage2": half, "stage3": m}

    return pyramid(projs), pyramid(intris)


def read_image_u8(filename):
    """cas_normal_eval.py:38-40, 114-115: `Image.open(filename)` -> `np.array(image)`; decoding stays on the host (PIL).
    Returns a contiguous uint8 [h,w,C] array, ready for `torch.from_numpy(...).cuda()` and center_image."""
    from PIL import Image

    img = np.array(Image.open(filename))
    if img.ndim == 2:
        img = img[:, :, None]
    if img.dtype != np.uint8:
        raise TypeError("%s decodes to %s; the device item builder takes 8-bit images" % (filename, img.dtype))
    return np.ascontiguousarray(img)


# ----------------------------------------------------------------------------------------
# crop + normalise on the device
# ----------------------------------------------------------------------------------------
_MODES = {"standard": 0, "mean": 1, "vit": 2}


def center_image(img_u8, mode="mean", window=None, out=None):
    """preprocess.py:92-117 center_image over the crop window of preprocess.py:76-80.
    img_u8: device uint8 tensor [h,w,C] (a decoded image, interleaved); window = (y0, x0, H, W) from slice_window, or
    None for the whole image.  Returns float32 [C,new_h,new_w] -- already in the layout the reference reaches with
    np.stack(...).transpose([0,3,1,2]) (cas_normal_eval.py:147)."""
    if mode not in _MODES:
        raise Exception("{}? Not implemented yet!".format(mode))
    if not (isinstance(img_u8, torch.Tensor) and img_u8.is_cuda):
        raise RuntimeError("img_u8 must be on the GPU (no CPU fallback)")
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 3 or not img_u8.is_contiguous():
        raise TypeError("img_u8 must be a contiguous uint8 [h,w,C] tensor")
    h, w, C = img_u8.shape
    y0, x0, H, W = window if window is not None else (0, 0, h, w)
    if out is None:
        out = torch.empty((C, H, W), dtype=torch.float32, device=img_u8.device)
    sums = torch.empty(8, dtype=torch.int64, device=img_u8.device)
    rc = _lib.load().d3d_center_image_u8(ctypes.c_void_p(img_u8.data_ptr()), h, w, C, int(y0), int(x0), int(H), int(W),
                                         _MODES[mode], ctypes.c_void_p(sums.data_ptr()), _chk(out, "out"), _stream())
    _lib.check(rc, "d3d_center_image_u8")
    return out


# ----------------------------------------------------------------------------------------
# feature pyramids kept resident across reference views
# ----------------------------------------------------------------------------------------
class FeatureCache(object):
    """LRU cache image key -> feature pyramid (dict stage -> [1,C,h,w] device tensor) bounded by `max_bytes` of HBM.

    A key must identify the image AND its preprocessing (path, crop window, normalisation mode); the cache belongs to
    one set of feature-network weights: call clear() after loading a checkpoint."""

    def __init__(self, max_bytes=64 << 30, by_content=False):
        """by_content=True: views that arrive WITHOUT keys (the reference's own item layout, `forward(imgs, ...)`
        unchanged) are recognised by their pixels -- a cheap device fingerprint selects a candidate and an exact
        comparison with the stored image confirms it, so a reused pyramid always belongs to a bit-identical input."""
        self.by_content = bool(by_content)
        self.max_bytes = int(max_bytes)
        self.bytes = 0
        self.hits = 0
        self.misses = 0
        self._d = collections.OrderedDict()

    @staticmethod
    def _size(pyr):
        return sum(t.numel() * t.element_size() for t in pyr.values())

    def get(self, key):
        pyr = self._d.get(key)
        if pyr is None:
            self.misses += 1
            return None
        self._d.move_to_end(key)
        self.hits += 1
        return pyr

    def put(self, key, pyr):
        size = self._size(pyr)
        if size > self.max_bytes:
            return
        if key in self._d:
            self.bytes -= self._size(self._d.pop(key))
        while self._d and self.bytes + size > self.max_bytes:
            _, old = self._d.popitem(last=False)
            self.bytes -= self._size(old)
        self._d[key] = pyr
        self.bytes += size

    def clear(self):
        self._d.clear()
        self.bytes = 0

    def __len__(self):
        return len(self._d)

    def __contains__(self, key):
        return key in self._d


def _by_content(feature_net, x, cache):
    """Pyramid of image tensor x [B,3,H,W], reused when a bit-identical tensor has been featurised before.  The
    fingerprint (two float64 reductions, one host read) only selects the candidate; torch.equal on the stored copy of
    the image decides.  The stored image counts against the cache budget (it rides in the entry as "_image")."""
    flat = x.reshape(-1)
    ramp = torch.arange(flat.numel() % 8191 + 1, dtype=torch.float64, device=x.device)  # short ramp, tiled by viewing
    d = flat.double()
    n = (flat.numel() // ramp.numel()) * ramp.numel()
    fp = torch.stack([d.sum(), (d[:n].view(-1, ramp.numel()) * ramp).sum()]).tolist()
    key = ("content", tuple(x.shape), fp[0], fp[1])
    entry = cache.get(key)
    if entry is not None and torch.equal(entry["_image"], x):
        return {k: v for k, v in entry.items() if k != "_image"}
    pyr = feature_net(x)
    stored = dict(pyr)
    stored["_image"] = x.clone()
    cache.put(key, stored)
    return pyr


def _pyramids(feature_net, xs):
    """feature_net(x) for every [B,3,H,W] tensor of xs.  On the GPU the images go round-robin over the caller's stream and two side
    streams (ops.on_streams: the pyramids are independent; the half- and quarter-resolution layers of one image leave most of the
    chip idle) and the caller's stream waits for the side streams before anything reads a pyramid.  Same kernels, same operands."""
    from . import ops

    return ops.on_streams([(lambda x=x: feature_net(x)) for x in xs], xs[0].device, "fpn_streams")


def extract_features(feature_net, imgs, image_keys=None, cache=None):
    """`[self.feature(imgs[:, v]) for v in range(V)]` (cas_mvsnet.py:189-192, adamvs.py:571-574, msrednet.py:
    482-485) with the pyramids of known images taken from `cache`.  imgs: [B,V,3,H,W] tensor, or a list of V entries
    each either a [B,3,H,W] tensor or None (None = "not uploaded because its key is cached")."""
    V = imgs.shape[1] if isinstance(imgs, torch.Tensor) else len(imgs)
    view = (lambda v: imgs[:, v]) if isinstance(imgs, torch.Tensor) else (lambda v: imgs[v])
    if cache is not None and image_keys is None and cache.by_content:
        return [_by_content(feature_net, view(v), cache) for v in range(V)]
    if cache is None or image_keys is None:
        return _pyramids(feature_net, [view(v) for v in range(V)])
    if len(image_keys) != V:
        raise ValueError("image_keys must name the %d views" % V)
    # two passes: take (and so hold a reference to) every cached pyramid of the item FIRST; a put() for an uncached view
    # may evict any key, including one of this item's that the caller did not upload because it was cached
    feats = [cache.get(k) for k in image_keys]
    # the misses of the item -- each image once (padded source lists repeat an image) -- are featurised TOGETHER on the three
    # streams of _pyramids, as an item without a cache is (the first view of a strip misses all V), then stored
    first = {}
    for v in range(V):
        if feats[v] is None and image_keys[v] not in first:
            if view(v) is None:
                raise KeyError("image %r is neither cached nor supplied" % (image_keys[v],))
            first[image_keys[v]] = v
    if first:
        order = sorted(first.values())
        pyrs = _pyramids(feature_net, [view(v) for v in order])
        for v, pyr in zip(order, pyrs):
            feats[v] = pyr
        for v in range(V):
            if feats[v] is None:
                feats[v] = feats[first[image_keys[v]]]
        for v in order:
            cache.put(image_keys[v], feats[v])
    return feats


# ----------------------------------------------------------------------------------------
# the block on disk: viewpair.txt / images.txt / cameras.txt / image_path.txt
# (mvs/mvs_cas/datasets/data_io.py:18-126 define the records and the four text formats;
#  mvs/mvs_cas/datasets/cas_normal_eval.py:10-182 is the dataset built on them)
# ----------------------------------------------------------------------------------------
class Camera(object):
    """One line of cameras.txt (data_io.py:18-29, 48-67): id  width height  pixelsize  fx fy x0 y0  [distortion...]."""

    def __init__(self, camera_id=None, size=None, pixelsize=None, focallength=None, x0y0=None, distortion=None):
        self.camera_id, self.size, self.pixelsize = camera_id, size, pixelsize
        self.focallength, self.x0y0, self.distortion = focallength, x0y0, distortion


class Photo(object):
    """One line of images.txt (data_io.py:32-45, 70-91): id camera_id  R (9, row-major, Rwc XrightYup)  C (3, twc)
    depth_min depth_max  name."""

    def __init__(self, image_id=None, camera_id=None, rotation_matrix=None, project_center=None, depth=None, name=None):
        self.image_id, self.camera_id, self.name = image_id, camera_id, name
        self.rotation_matrix, self.project_center, self.depth = rotation_matrix, project_center, depth


def _records(path):
    with open(path, "r") as f:
        for line in f:
            line = line.strip()
            if line and not line.startswith("#"):
                yield line.split()


def read_cameras_text(path):
    """data_io.py:48-67 -> {camera_id: Camera}."""
    cams = {}
    for e in _records(path):
        p = np.array(tuple(map(float, e[4:8])))
        cams[int(e[0])] = Camera(camera_id=int(e[0]), size=[int(e[1]), int(e[2])], pixelsize=float(e[3]),
                                 focallength=[p[0], p[1]], x0y0=[p[2], p[3]],
                                 distortion=np.array(tuple(map(float, e[8:]))))
    return cams


def read_images_text(path):
    """data_io.py:70-91 -> {image_id: Photo}."""
    images = {}
    for e in _records(path):
        images[int(e[0])] = Photo(image_id=int(e[0]), camera_id=int(e[1]),
                                  rotation_matrix=np.array(tuple(map(float, e[2:11]))).reshape(3, 3),
                                  project_center=np.array(tuple(map(float, e[11:14]))),
                                  depth=np.array(tuple(map(float, e[14:16]))), name=e[16])
    return images


def read_images_path_text(path):
    """data_io.py:94-108: `N` then N triples `index name path` (whitespace separated) -> ({index: path}, {index: name})."""
    tok = open(path).read().split()
    paths, names = {}, {}
    for i in range(int(tok[0])):
        idx = int(tok[3 * i + 1])
        names[idx], paths[idx] = tok[3 * i + 2], tok[3 * i + 3]
    return paths, names


def read_view_pair_text(pair_path, view_num):
    """data_io.py:111-126: per viewpoint a line with the reference id and a line `n id score id score ...`; views with no
    source are dropped, short source lists are padded with their first source.  -> [[ref, src...], ...] (every source
    listed is kept; an item uses the first view_num entries, cas_normal_eval.py:105-107)."""
    metas = []
    with open(pair_path) as f:
        for _ in range(int(f.readline())):
            ref = [int(f.readline().rstrip())]
            src = [int(x) for x in f.readline().rstrip().split()[1::2]]
            if src:
                if len(src) < view_num:
                    print("{}< num_views:{}".format(len(src), view_num))
                    src += [src[0]] * (view_num - len(src))
                metas.append(ref + src)
    return metas


def read_scene_blocks(path):
    """blocks.txt (IO/params_io.py:430-444, read by fuse/fusion_3d_normal.py:252-272): `N`, then per scene block a line with its
    range `x_min x_max y_min y_max z_min z_max` and a line with the image ids of its reference views.
    -> [{"scene_range": [6 floats], "refs": [image ids]}]."""
    blocks = []
    with open(path) as f:
        for _ in range(int(f.readline())):
            rng = [float(x) for x in f.readline().split()]
            refs = [int(x) for x in f.readline().split()]
            blocks.append({"scene_range": rng, "refs": refs})
    return blocks


def scale_image(image, scale=1.0):
    """preprocess.py:41-46 scale_image -> cv2.resize(image, None, fx=scale, fy=scale, INTER_LINEAR).
    scale == 1 (the pipeline's setting: mvs_dl.py never passes --resize_scale) returns the image as cv2 does.  Other
    scales follow OpenCV's documented INTER_LINEAR rule (output size round(n*scale); source coordinate
    (dst + 0.5)/scale - 0.5 with the GIVEN scale -- not n_in/n_out --, clamped; round half up for 8-bit) in float64 -- OpenCV itself is absent from the build
    image, so this branch is NOT pinned against it (its 8-bit path uses 11-bit fixed-point weights and may differ by
    one grey level)."""
    if scale == 1 or scale == 1.0:
        return np.ascontiguousarray(image)
    img = np.asarray(image)
    h, w = img.shape[:2]
    nh, nw = int(round(h * scale)), int(round(w * scale))

    def taps(n_out, n_in, s):
        x = (np.arange(n_out) + 0.5) / s - 0.5
        x0 = np.floor(x).astype(np.int64)
        f = np.where((x0 < 0) | (x0 >= n_in - 1), 0.0, x - x0)   # OpenCV zeroes the fraction where a tap is clamped
        a = np.clip(x0, 0, n_in - 1)
        b = np.clip(x0 + 1, 0, n_in - 1)
        return a, b, f

    ya, yb, fy = taps(nh, h, float(scale))   # cv2.resize(..., fx=scale, fy=scale) keeps inv_scale = 1/scale, NOT n_in/n_out
    xa, xb, fx = taps(nw, w, float(scale))
    src = img.astype(np.float64)
    if src.ndim == 2:
        src = src[:, :, None]
    top = src[ya][:, xa] * (1 - fx)[None, :, None] + src[ya][:, xb] * fx[None, :, None]
    bot = src[yb][:, xa] * (1 - fx)[None, :, None] + src[yb][:, xb] * fx[None, :, None]
    out = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
    if img.dtype == np.uint8:
        out = np.clip(np.floor(out + 0.5), 0, 255).astype(np.uint8)
    else:
        out = out.astype(img.dtype)
    return out.reshape((nh, nw) + img.shape[2:])


def create_cams(image_params, cam_params_dict, num_depth=384, min_interval=0.1):
    """cas_normal_eval.py:53-91: images.txt holds [Rwc | twc] with the camera looking along -z, y up; the network wants
    Tcw with x right, y down.  cam[0] = inverse([Rwc @ diag(1,-1,-1) | twc]) (inverted in float32, as the reference does),
    cam[1][:3,:3] = K, cam[1][3] = (depth_min, (depth_max - depth_min) / num_depth, num_depth, depth_max).
    `min_interval` is accepted and unused, as in the reference."""
    cam = np.zeros((2, 4, 4), dtype=np.float32)
    extrinsics = np.zeros((4, 4), dtype=np.float32)
    flip_yz = np.array([[1, 0, 0], [0, -1, 0], [0, 0, -1]], dtype=float)
    extrinsics[0:3, 0:3] = np.matmul(image_params.rotation_matrix, flip_yz)
    extrinsics[0:3, 3] = image_params.project_center
    extrinsics[3, 3] = 1.0
    cam[0, :, :] = np.linalg.inv(extrinsics)
    cp = cam_params_dict[image_params.camera_id]
    cam[1][0][0], cam[1][1][1] = cp.focallength[0], cp.focallength[1]
    cam[1][0][2], cam[1][1][2] = cp.x0y0[0], cp.x0y0[1]
    cam[1][2][2] = 1
    cam[1][3][0] = image_params.depth[0]
    cam[1][3][1] = (image_params.depth[1] - image_params.depth[0]) / num_depth
    cam[1][3][3] = image_params.depth[1]
    cam[1][3][2] = num_depth
    return cam


def _host_center_image(img, mode="mean"):
    """preprocess.py:92-115 on the host (the reference's item layout carries normalised float images)."""
    if mode == "standard":
        return np.array(img, dtype=np.float32) / 255.
    if mode == "mean":
        x = np.array(img).astype(np.float32)
        var = np.var(x, axis=(0, 1), keepdims=True)
        mean = np.mean(x, axis=(0, 1), keepdims=True)
        return (x - mean) / (np.sqrt(var) + 0.00000001)
    if mode == "vit":
        x = np.array(img).astype(np.float32)
        return (x - np.array([123.675, 116.28, 103.53], np.float32)) / (np.array([58.395, 57.12, 57.375], np.float32) + 0.00000001)
    raise Exception("{}? Not implemented yet!".format(mode))


class MVSDataset(object):
    """The inference dataset of the reference (cas_normal_eval.py:10-182) over a block folder holding viewpair.txt,
    images.txt, cameras.txt and image_path.txt.  Same constructor, same `len`, and `dataset[i]` returns the same item
    dict (imgs [V,3,H,W] float32 normalised on the host, proj_matrices / intri_matrices {stage1..3}, depth_values
    [min, max], outimage, outcam, ref_image_path, outlocation).

    `device_item(i)` is the form the predict loop of this package uses: decoded 8-bit images with their crop windows and
    cache keys ("images_u8", "crop_windows", "image_keys") -- cropping and normalisation then run on the GPU
    (center_image) and an image shared by several reference views is featurised once (FeatureCache); the matrices and
    the output records are the same objects as in `dataset[i]`.

    args needs: min_interval, interval_scale, numdepth, resize_scale, sample_scale, max_h, max_w (predict.py:38-48)."""

    def __init__(self, data_folder, mode, view_num, normalize, args, **kwargs):
        assert mode in ["train", "val", "test"]
        self.data_folder, self.mode, self.args = data_folder, mode, args
        self.view_num, self.normalize = view_num, normalize
        self.min_interval, self.interval_scale, self.num_depth = args.min_interval, args.interval_scale, args.numdepth
        self.cam_params_dict = read_cameras_text(data_folder + "/cameras.txt")
        self.image_params_dict = read_images_text(data_folder + "/images.txt")
        self.image_paths, _ = read_images_path_text(data_folder + "/image_path.txt")
        self.sample_list = read_view_pair_text(data_folder + "/viewpair.txt", view_num)
        self.sample_num = len(self.sample_list)

    def __len__(self):
        return len(self.sample_list)

    def _view(self, image_idx):
        """Decoded image, its (scaled, cropped) camera and crop window: cas_normal_eval.py:112-127."""
        image = read_image_u8(self.image_paths[image_idx])
        ip = self.image_params_dict[image_idx]
        cam = create_cams(ip, self.cam_params_dict, self.num_depth, self.min_interval * self.interval_scale)
        rs = self.args.resize_scale
        image = scale_image(image, rs)
        cam = scale_camera(cam, scale=rs)
        win = crop_window(image.shape[0], image.shape[1], self.args.max_h, self.args.max_w, resize_scale=rs)
        cam = crop_camera(cam, win[0], win[1])
        return image, cam, win, ip

    def _records_of(self, views):
        cams = [v[1] for v in views]
        image0, cam0, win0, ip0 = views[0]
        y0, x0, H, W = slice_window(image0.shape[0], image0.shape[1], win0)
        pm, im = stage_projections(cams, sample_scale=self.args.sample_scale)
        depth_values = np.array([cam0[1][3][0], cam0[1][3][3]], dtype=np.float32)
        return {"proj_matrices": pm, "intri_matrices": im, "depth_values": depth_values, "outcam": cam0,
                "outlocation": [str(W), str(H), str(ip0.image_id), str(ip0.name)]}

    def __getitem__(self, idx):
        ids = self.sample_list[idx][:self.view_num]
        views = [self._view(i) for i in ids]
        item = self._records_of(views)
        crops = []
        for image, _, win, _ in views:
            y0, x0, H, W = slice_window(image.shape[0], image.shape[1], win)
            crops.append(image[y0:y0 + H, x0:x0 + W])
        item["imgs"] = np.stack([_host_center_image(c, self.normalize) for c in crops]).transpose([0, 3, 1, 2])
        item["outimage"] = crops[0]
        item["ref_image_path"] = self.image_paths[ids[0]]
        return item

    def view_records(self, fusion_num=10):
        """The view list the fusion step works from (fuse/fusion_3d_normal.py:98-99, 227-249: viewpair.txt read again with
        fusion_num sources -- not predict's view_num): per item {"name": the reference view's product name, "src": the names of
        ALL its listed sources up to fusion_num, "id": the 1-based position of the image in the block's image list (what
        Fuse_Depth_Map.read_ImageID counts, :284-303; 0 means "not visible" in the visibility planes)}.  No image is read."""
        name = lambda i: os.path.splitext(str(self.image_params_dict[i].name))[0]
        recs = []
        for s in self.sample_list:
            src = list(s[1:])
            if len(src) < fusion_num:   # :241-244: short lists are filled up with their first source (it is then checked again)
                src += [src[0]] * (fusion_num - len(src))
            recs.append({"name": name(s[0]), "src": [name(j) for j in src[:fusion_num]], "id": int(s[0]) + 1, "image": int(s[0])})
        return recs

    def device_item(self, idx):
        ids = self.sample_list[idx][:self.view_num]
        views = [self._view(i) for i in ids]
        item = self._records_of(views)
        wins = [slice_window(v[0].shape[0], v[0].shape[1], v[2]) for v in views]
        item["images_u8"] = [v[0] for v in views]
        item["crop_windows"] = wins
        item["normalize"] = self.normalize
        item["image_keys"] = [(self.image_paths[i], self.args.resize_scale, w, self.normalize) for i, w in zip(ids, wins)]
        item["ref_image_path"] = self.image_paths[ids[0]]
        return item


class DeviceItems(object):
    """`dataset[i]` -> `dataset.device_item(i)` view of an MVSDataset, for predict_views."""

    def __init__(self, dataset):
        self.dataset = dataset

    def __len__(self):
        return len(self.dataset)

    def __getitem__(self, idx):
        return self.dataset.device_item(idx)

    def view_records(self, fusion_num=10):
        return self.dataset.view_records(fusion_num)
