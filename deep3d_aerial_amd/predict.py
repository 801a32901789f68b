"""Inference harness with the reference's predict.py boundary (mvs/mvs_cas/predict.py).

What is kept identical for the rest of the pipeline (mvs_dl.py upstream, fuse/ downstream):
  * the CLI flags mvs_dl.py formats (predict.py:30-58, mvs/mvs_dl.py:61-63);
  * the model switch and its error for unknown names (predict.py:71-97);
  * checkpoint layout {'model': state_dict} with DataParallel 'module.' prefixes (predict.py:105-106);
  * the products per reference view: {name}_init.pfm, {name}_prob.pfm, {name}.txt
    (predict.py:146-183; PFM layout data_io.py:196-223; camera text data_io.py:291-314).
What is new: reference views are sharded over ranks (one process per GPU, sharding.py).

The block folder (viewpair.txt, images.txt, cameras.txt, image_path.txt) is read by dataset.MVSDataset, the
counterpart of datasets/cas_normal_eval.py; any iterable yielding the same sample dicts can be passed to
predict_views(), and SyntheticBlock provides one for plumbing tests.
"""
import argparse
import os
import sys

import numpy as np
import torch

from . import sharding, synthetic


# ----------------------------------------------------------------------------------------
# output files
# ----------------------------------------------------------------------------------------
def save_pfm(filename, image, scale=1):
    """data_io.py:196-223 save_pfm_utf8: 'Pf', 'W H', scale (negative = little endian), rows bottom-up."""
    image = np.asarray(image)
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    if image.ndim == 3 and image.shape[2] == 3:
        color = True
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        color = False
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    image = np.flipud(image)
    endian = image.dtype.byteorder
    if endian == "<" or (endian == "=" and sys.byteorder == "little"):
        scale = -scale
    with open(filename, "wb") as f:
        f.write(b"PF\n" if color else b"Pf\n")
        f.write(("%d %d\n" % (image.shape[1], image.shape[0])).encode("utf-8"))
        f.write(("%f\n" % scale).encode("utf-8"))
        image.tofile(f)


def load_pfm(filename):
    with open(filename, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        if header not in ("PF", "Pf"):
            raise Exception("Not a PFM file.")
        w, h = (int(t) for t in f.readline().decode("utf-8").split())
        scale = float(f.readline().decode("utf-8").rstrip())
        data = np.fromfile(f, "<f4" if scale < 0 else ">f4")
    shape = (h, w, 3) if header == "PF" else (h, w)
    return np.flipud(data.reshape(shape)).astype(np.float32), abs(scale)


def _pfm_header(h, w, color=False, scale=1):
    # little-endian float32 payload => negative scale (data_io.py:215-219)
    return (b"PF\n" if color else b"Pf\n") + ("%d %d\n" % (w, h)).encode("utf-8") + ("%f\n" % -scale).encode("utf-8")


class PfmWriter(object):
    """Asynchronous writer of the per-view products (predict.py:176-180: {name}_init.pfm, {name}_prob.pfm) --
    SURVEY.md 8f row N2.  The reference pulls each map to the host synchronously, flips it with NumPy and writes it
    before the next view starts.  Here `submit` (stream-ordered, returns at once) flips the maps on the device
    into a staging buffer in file order (`d3d_flip_rows`), a copy stream moves the buffer to pinned host memory with
    ONE asynchronous D2H transfer, and a writer thread puts header + payload on disk while the GPU runs the next
    view.  `depth` slots bound the memory in flight.  Files are byte-identical to save_pfm's."""

    def __init__(self, h, w, n_maps=2, depth=2, device="cuda"):
        import queue
        import threading

        if not torch.cuda.is_available():
            raise RuntimeError("PfmWriter stages through the GPU (no CPU fallback); use save_pfm for host arrays")
        self.h, self.w, self.n = int(h), int(w), int(n_maps)
        self.device = torch.device(device)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.slots = [dict(dev=torch.empty((self.n, self.h, self.w), dtype=torch.float32, device=self.device),
                           host=torch.empty((self.n, self.h, self.w), dtype=torch.float32).pin_memory(),
                           free=threading.Event()) for _ in range(depth)]
        for sl in self.slots:
            sl["free"].set()
        self._next = 0
        self._q = queue.Queue()
        self._err = None
        self._thread = threading.Thread(target=self._run, name="pfm-writer", daemon=True)
        self._thread.start()

    def submit(self, maps, paths, display=None):
        """maps: n device tensors of h*w fp32 elements each; paths: n file names.  display = (output_folder, name) has the
        writer thread render the colour maps of predict.py:155-176 from the same host copy (maps[0] depth, maps[1]
        confidence) after the files are written -- the GPU does not wait for the PNG encoder."""
        import ctypes

        from . import _lib
        from .ops import _chk, _stream

        if self._err is not None:
            raise self._err
        if len(maps) != self.n or len(paths) != self.n:
            raise ValueError("expected %d maps and paths" % self.n)
        maps = [m.reshape(self.h, self.w) for m in maps]
        sl = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        sl["free"].wait()
        sl["free"].clear()
        ptrs = (ctypes.c_void_p * self.n)(*[_chk(m, "map").value for m in maps])
        _lib.check(_lib.load().d3d_flip_rows(ptrs, self.n, self.h, self.w, _chk(sl["dev"], "staging"), _stream()),
                   "d3d_flip_rows")
        cur = torch.cuda.current_stream(self.device)
        self.copy_stream.wait_stream(cur)
        with torch.cuda.stream(self.copy_stream):
            sl["host"].copy_(sl["dev"], non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        self._q.put((sl, done, list(paths), display))

    def _run(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            sl, done, paths, display = item
            try:
                done.synchronize()
                payload = sl["host"].numpy()
                for k, path in enumerate(paths):
                    with open(path, "wb") as f:
                        f.write(_pfm_header(self.h, self.w))
                        payload[k].tofile(f)
                if display is not None:   # the staging buffer is in file order (bottom row first): flip back
                    write_display_maps(display[0], display[1], payload[0][::-1].copy(), payload[1][::-1].copy())
            except Exception as e:  # surfaced by the next submit() / close()
                self._err = e
            finally:
                sl["free"].set()

    def close(self):
        """Waits until every submitted file is on disk."""
        self._q.put(None)
        self._thread.join()
        if self._err is not None:
            raise self._err

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


def load_pfm_device(filename, device="cuda"):
    """Read side for the fusion step (fusion_3d_normal.py:444,488: read_pfm(path)[0]): the payload goes to the
    device as it lies in the file (pinned host buffer, one H2D copy) and is un-flipped there.  Greyscale
    little-endian files (what the writer produces); anything else takes load_pfm + upload."""
    import ctypes

    from . import _lib
    from .ops import _chk, _stream

    with open(filename, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        w, h = (int(t) for t in f.readline().decode("utf-8").split())
        scale = float(f.readline().decode("utf-8").rstrip())
        if header != "Pf" or scale >= 0 or sys.byteorder != "little":
            return torch.from_numpy(np.ascontiguousarray(load_pfm(filename)[0])).to(device)
        host = torch.empty((h, w), dtype=torch.float32).pin_memory()
        n = f.readinto(memoryview(host.numpy()).cast("B"))
        if n != h * w * 4:
            raise Exception("truncated PFM payload in %s" % filename)
    raw = host.to(device, non_blocking=True)
    out = torch.empty_like(raw)
    ptrs = (ctypes.c_void_p * 1)(_chk(raw, "payload").value)
    _lib.check(_lib.load().d3d_flip_rows(ptrs, 1, h, w, _chk(out, "out"), _stream()), "d3d_flip_rows")
    torch.cuda.current_stream().synchronize()  # the pinned buffer is released on return
    return out


def write_red_cam(file, cam, location, ref_path):
    """data_io.py:291-314: extrinsic 4x4, intrinsic 3x3, 'dmin interval D dmax', location + ref path."""
    with open(file, "w") as f:
        f.write("extrinsic: XrightYdown, [Rcw|tcw]\n")
        for i in range(4):
            for j in range(4):
                f.write(str(cam[0][i][j]) + " ")
            f.write("\n")
        f.write("\n")
        f.write("intrinsic\n")
        for i in range(3):
            for j in range(3):
                f.write(str(cam[1][i][j]) + " ")
            f.write("\n")
        f.write("\n" + str(cam[1][3][0]) + " " + str(cam[1][3][1]) + " " + str(cam[1][3][2]) + " " +
                str(cam[1][3][3]) + "\n")
        f.write("\n")
        for word in location:
            f.write(str(word) + " ")
        f.write(str(ref_path) + "\n")


# ----------------------------------------------------------------------------------------
# model switch (predict.py:71-97) and checkpoint loading (predict.py:105-106)
# ----------------------------------------------------------------------------------------
def build_model(name, numdepth, ndepths=(48, 32, 8), depth_inter_r=(4, 2, 1), share_cr=False,
                cr_base_chs=(8, 8, 8)):
    if name == "casmvsnet":
        from .cas_mvsnet import Infer_CascadeMVSNet
        return Infer_CascadeMVSNet(num_depth=numdepth, ndepths=list(ndepths),
                                   depth_intervals_ratio=list(depth_inter_r), share_cr=share_cr,
                                   cr_base_chs=list(cr_base_chs))
    if name == "adamvs":
        from .adamvs import Infer_AdaMVSNet
        return Infer_AdaMVSNet(num_depth=numdepth, ndepths=list(ndepths), depth_intervals_ratio=list(depth_inter_r),
                               share_cr=share_cr, cr_base_chs=list(cr_base_chs))
    if name == "msrednet":
        from .msrednet import Infer_CascadeREDNet
        return Infer_CascadeREDNet(num_depth=numdepth, ndepths=list(ndepths),
                                   depth_intervals_ratio=list(depth_inter_r), share_cr=share_cr,
                                   cr_base_chs=list(cr_base_chs))
    if name == "ucsnet":
        # the reference's own call (predict.py:78-81) fails in the constructor (no `num_depth` argument, and
        # forward reads self.num_depth); here the class takes it, so `--model ucsnet` runs
        from .ucsnet import Infer_UCSNet
        return Infer_UCSNet(lamb=1.5, num_depth=numdepth, ndepths=list(ndepths))
    raise Exception("{}? Not implemented yet!".format(name))


def load_checkpoint(model, path):
    """Accepts the reference's {'model': sd} files, with or without DataParallel's 'module.' prefix."""
    sd = torch.load(path, map_location="cpu")
    sd = sd["model"] if "model" in sd else sd
    sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
    model.load_state_dict(sd)
    return model


# ----------------------------------------------------------------------------------------
# a synthetic block with the dataset item layout of cas_normal_eval.py:94-182
# ----------------------------------------------------------------------------------------
class SyntheticBlock:
    def __init__(self, n_items, view_num, max_h, max_w, numdepth, seed=0):
        self.n, self.v, self.h, self.w, self.nd, self.seed = n_items, view_num, max_h, max_w, numdepth, seed

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        rng = np.random.default_rng(self.seed + idx)
        imgs = rng.standard_normal((self.v, 3, self.h, self.w), dtype=np.float32)
        proj, dv = synthetic.make_scene(self.v, self.h, self.w, self.nd, sweep_px=min(self.nd, self.w) / 4.0,
                                        seed=self.seed + idx)
        pm = {}
        for name, sc in (("stage1", 0.25), ("stage2", 0.5), ("stage3", 1.0)):
            q = proj.copy()
            q[:, :2, :] = proj[:, :2, :] * np.float32(sc)
            pm[name] = q
        cam = np.zeros((2, 4, 4), np.float32)
        cam[0] = np.eye(4)
        cam[1, :3, :3] = np.array([[1.2 * max(self.h, self.w), 0, (self.w - 1) / 2],
                                   [0, 1.2 * max(self.h, self.w), (self.h - 1) / 2], [0, 0, 1]])
        cam[1, 3] = [dv[0], (dv[1] - dv[0]) / self.nd, self.nd, dv[1]]
        name = "view_%04d" % idx
        return {"imgs": imgs, "proj_matrices": pm, "depth_values": dv, "outcam": cam,
                "outlocation": [str(self.w), str(self.h), str(idx), name + ".png"], "ref_image_path": name + ".png"}


class SyntheticStrip:
    """A flight strip of `n_images` overlapping 8-bit images; reference view i uses images i, i+1, ... (ring), as
    neighbouring reference views of a real block share most of their sources (viewpair.txt).  Items carry what the
    reference's item builder starts from -- decoded images and [2,4,4] cameras (cas_normal_eval.py:112-127) -- so the
    crop / normalise / projection steps run through dataset.py on the GPU, and `image_keys` name the shared images
    for the feature cache.  Images are a little larger than (max_h, max_w): the centre crop is exercised."""

    def __init__(self, n_images, view_num, max_h, max_w, numdepth, seed=0, border=(6, 10), normalize="mean"):
        self.n, self.v, self.h, self.w, self.nd, self.seed = n_images, view_num, max_h, max_w, numdepth, seed
        self.h0, self.w0 = max_h + border[0], max_w + border[1]
        self.normalize = normalize
        self.depth = (400.0, 800.0)

    def __len__(self):
        return self.n

    def image(self, j):
        rng = np.random.default_rng(self.seed * 1000 + j)
        base = rng.integers(0, 256, (self.h0 // 4 + 2, self.w0 // 4 + 2, 3), dtype=np.uint8)
        img = np.repeat(np.repeat(base, 4, axis=0), 4, axis=1)[:self.h0, :self.w0]
        return np.ascontiguousarray(img ^ rng.integers(0, 16, img.shape, dtype=np.uint8))

    def camera(self, j):
        from . import dataset

        f = 1.2 * max(self.h0, self.w0)
        cam = np.zeros((2, 4, 4), np.float32)
        R = synthetic._rot(0.004 * np.sin(j), 0.006 * np.cos(1.3 * j), 0.003 * j)
        C = np.array([0.02 * self.depth[0] * j, 0.004 * self.depth[0] * np.sin(j), 0.0])
        cam[0] = np.eye(4)
        cam[0, :3, :3] = R
        cam[0, :3, 3] = -R @ C
        cam[1, :3, :3] = [[f, 0, (self.w0 - 1) / 2], [0, f, (self.h0 - 1) / 2], [0, 0, 1]]
        cam[1, 3] = [self.depth[0], (self.depth[1] - self.depth[0]) / self.nd, self.nd, self.depth[1]]
        win = dataset.crop_window(self.h0, self.w0, self.h, self.w)
        return dataset.crop_camera(cam, win[0], win[1]), dataset.slice_window(self.h0, self.w0, win)

    def view_records(self, fusion_num=10):
        """pipeline.view_records: reference view i lists images i + 1, i + 2, ... (the ring) as its sources."""
        return [{"name": "view_%04d" % i, "src": ["view_%04d" % ((i + k) % self.n) for k in range(1, min(self.n, 1 + fusion_num))],
                 "id": i + 1, "image": i} for i in range(self.n)]

    def __getitem__(self, idx):
        from . import dataset

        ids = [(idx + k) % self.n for k in range(self.v)]
        cams, wins = zip(*[self.camera(j) for j in ids])
        pm, _ = dataset.stage_projections(list(cams))
        name = "view_%04d" % idx
        return {"images_u8": [self.image(j) for j in ids], "image_keys": [(self.seed, j, w, self.normalize) for j, w in zip(ids, wins)],
                "crop_windows": list(wins), "normalize": self.normalize, "proj_matrices": pm,
                "depth_values": np.array(self.depth, np.float32), "outcam": cams[0],
                "outlocation": [str(self.w), str(self.h), str(idx), name + ".png"], "ref_image_path": name + ".png"}


def _item_views(s, model, device):
    """The views of one dataset item on the device.  Items with decoded 8-bit images ("images_u8", "crop_windows",
    "image_keys") are cropped and normalised on the GPU (dataset.center_image), and an image whose feature pyramid is
    cached is not even uploaded; items with host-normalised float images ("imgs", the reference's layout) are uploaded
    as they are."""
    from . import dataset

    if "images_u8" not in s:
        return torch.from_numpy(np.ascontiguousarray(s["imgs"]))[None].to(device), s.get("image_keys")
    cache = getattr(model, "feature_cache", None)
    views = []
    for v, im in enumerate(s["images_u8"]):
        if cache is not None and s["image_keys"][v] in cache:
            views.append(None)
            continue
        u8 = torch.from_numpy(np.ascontiguousarray(im)).to(device, non_blocking=True)
        views.append(dataset.center_image(u8, s.get("normalize", "mean"), s["crop_windows"][v])[None])
    return views, s["image_keys"]


# ----------------------------------------------------------------------------------------
# the per-view loop of predict.py:126-183, sharded over ranks
# ----------------------------------------------------------------------------------------
def predict_views(model, dataset, output_folder, rank=0, world_size=1, device="cuda", keep_maps=False,
                  feature_cache_bytes=0, display=False, partition="block", stats=None, cams=None):
    """Returns the names of the views this rank produced; with keep_maps=True a dict name -> (depth, confidence)
    of device tensors instead, so the fusion step (fuse.ViewFusion) can start without re-reading the PFM files.
    feature_cache_bytes > 0 keeps the feature pyramids of that many bytes of images resident across views (items must
    carry "image_keys"); results do not change.  partition: how the views are dealt to the ranks (sharding.shard_views;
    "block" keeps neighbouring views -- which share source images -- on one rank, so its cache keeps hitting).
    stats: a dict that receives this rank's view count, cache hits / misses and feature pyramids computed per view.
    cams: a dict that receives name -> the view's [2,4,4] camera (`outcam`: what write_red_cam puts into {name}.txt), for the
    fusion step that follows in the same process (pipeline.predict_and_fuse)."""
    from .dataset import FeatureCache

    os.makedirs(output_folder, exist_ok=True)
    if display:
        _display_backend()   # a missing matplotlib fails here, before the first view is computed
    model.eval()
    if feature_cache_bytes > 0:
        # items without "image_keys" (the reference's layout) are matched by content: exact comparison, same results
        model.feature_cache = FeatureCache(feature_cache_bytes, by_content=True)
    done = {} if keep_maps else []
    writer = None
    pyramids = []
    hook = model.feature.register_forward_hook(lambda *_: pyramids.append(1)) if stats is not None and hasattr(model, "feature") else None
    try:
        with torch.no_grad():
            for idx in sharding.shard_views(len(dataset), rank, world_size, partition):
                s = dataset[idx]
                imgs, keys = _item_views(s, model, device)
                pm = {k: torch.from_numpy(np.ascontiguousarray(v))[None].to(device)
                      for k, v in s["proj_matrices"].items()}
                dvh = np.ascontiguousarray(s["depth_values"])
                dv = torch.from_numpy(dvh)[None].to(device)
                from . import ops as _ops
                _ops.note_depth_range(dv, dvh.reshape(-1)[0], dvh.reshape(-1)[-1])   # the forward then never waits for the previous view
                out = model(imgs, pm, dv, image_keys=keys)
                depth = out["depth"].squeeze().float().contiguous()
                prob = out["photometric_confidence"].squeeze().float().contiguous()
                name = os.path.splitext(s["outlocation"][3])[0]
                paths = [os.path.join(output_folder, "%s_init.pfm" % name),
                         os.path.join(output_folder, "%s_prob.pfm" % name)]
                if writer is None or (writer.h, writer.w) != tuple(depth.shape):
                    if writer is not None:
                        writer.close()
                    writer = PfmWriter(depth.shape[0], depth.shape[1], 2, device=depth.device)
                # the files (and with --display the colour maps) land while the next view is computed
                writer.submit([depth, prob], paths, display=(output_folder, name) if display else None)
                write_red_cam(os.path.join(output_folder, "%s.txt" % name), s["outcam"], s["outlocation"],
                              s["ref_image_path"])
                if cams is not None:
                    cams[name] = np.array(s["outcam"], dtype=np.float32)
                if keep_maps:
                    done[name] = (depth, prob)
                else:
                    done.append(name)
    finally:
        if writer is not None:
            writer.close()
        if hook is not None:
            hook.remove()
        if stats is not None:
            cache = getattr(model, "feature_cache", None)
            n = len(done)
            stats.update(views=n, partition=partition, cache_hits=cache.hits if cache is not None else 0,
                         cache_misses=cache.misses if cache is not None else 0,
                         pyramids_per_view=(len(pyramids) / n) if n else 0.0)
        if feature_cache_bytes > 0:
            model.feature_cache = None
    return done


def _truthy(text):
    """--display arrives as the STRING "True" / "False" (mvs_dl.py:61-63 formats the bool; predict.py:50 declares no
    type), which the reference then tests for truth -- so its "False" displays too.  Here the
    text is read for what it says."""
    return str(text).strip().lower() not in ("", "0", "false", "no", "none")


def parse_args(argv=None):
    """Every flag of the reference's harness (predict.py:30-58) with its default, plus --synthetic_items /
    --random_weights / --feature_cache_gb of this package."""
    ap = argparse.ArgumentParser(description="plane-sweep depth inference (predict.py-compatible flags)")
    ap.add_argument("--model", default="adamvs", help="casmvsnet | msrednet | adamvs | ucsnet")
    ap.add_argument("--dataset", default="cas_normal_eval", help="dataset class (only the inference dataset exists here)")
    ap.add_argument("--data_folder", default=None, help="block folder: viewpair.txt images.txt cameras.txt image_path.txt")
    ap.add_argument("--output_folder", required=True)
    ap.add_argument("--loadckpt", default=None)
    ap.add_argument("--view_num", type=int, default=5)
    ap.add_argument("--numdepth", type=int, default=192)
    ap.add_argument("--max_w", type=int, default=3584)
    ap.add_argument("--max_h", type=int, default=4096)
    ap.add_argument("--min_interval", type=float, default=0.1)
    ap.add_argument("--fext", type=str, default=".jpg", help="accepted and unused, as in the reference's inference dataset")
    ap.add_argument("--normalize", type=str, default="mean")
    ap.add_argument("--resize_scale", type=float, default=1.0)
    ap.add_argument("--sample_scale", type=float, default=1)
    ap.add_argument("--interval_scale", type=float, default=1)
    ap.add_argument("--batch_size", type=int, default=1)
    ap.add_argument("--display", default=True)
    ap.add_argument("--share_cr", action="store_true")
    ap.add_argument("--ndepths", type=str, default="48,32,8")
    ap.add_argument("--depth_inter_r", type=str, default="4,2,1")
    ap.add_argument("--cr_base_chs", type=str, default="8,8,8")
    ap.add_argument("--synthetic_items", type=int, default=0, help="run on a synthetic block of this many views instead of --data_folder")
    ap.add_argument("--random_weights", action="store_true", help="run without --loadckpt (seeded random weights; plumbing tests only)")
    ap.add_argument("--feature_cache_gb", type=float, default=8.0, help="HBM kept for feature pyramids of shared images (0 = off)")
    ap.add_argument("--partition", default="block", choices=list(sharding.POLICIES),
                    help="views per rank: contiguous blocks (neighbouring views share images: the cache keeps hitting) or round_robin")
    # --fuse: BASELINE config 5's pipeline in one launch (pipeline.predict_and_fuse): the ranks all-gather their maps and every
    # rank fuses the reference views it swept.  Thresholds and counts are Fuse_Depth_Map's (fusion_3d_normal.py:56-57, run.py:176).
    ap.add_argument("--fuse", action="store_true", help="all-gather the depth / confidence maps and fuse this rank's reference views")
    ap.add_argument("--fusion_num", type=int, default=10)
    ap.add_argument("--geo_consist_num", type=int, default=4)
    ap.add_argument("--photometric_threshold", type=float, default=0.2)
    ap.add_argument("--position_threshold", type=float, default=1.0)
    ap.add_argument("--depth_threshold", type=float, default=0.01)
    ap.add_argument("--normal_threshold", type=float, default=90.0)
    ap.add_argument("--fuse_filter_sources", type=int, default=1,
                    help="1: confirmed samples leave the source maps (the reference's save_temp chain, kept per rank); 0: order-free")
    ap.add_argument("--fusion_output", default=None, help="folder of the fused arrays (default <output_folder>/fused)")
    ap.add_argument("--fuse_partition", default="views", choices=["views", "scene_blocks"],
                    help="who fuses what: every rank the reference views it swept (balanced; the filtering chain restarts at rank seams), or "
                         "whole scene blocks of blocks.txt per rank (the reference's chains exactly, for any number of ranks)")
    ap.add_argument("--blocks_file", default=None, help="scene blocks (default <data_folder>/blocks.txt) for --fuse_partition scene_blocks")
    return ap.parse_args(argv)


def _display_backend():
    import matplotlib

    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    return plt


def write_display_maps(output_folder, name, depth, prob):
    """predict.py:155-176: colour renderings of the depth (36000 - depth, non-finite columns patched) and confidence
    maps under <output>/color/.  Needs matplotlib, like the reference."""
    plt = _display_backend()
    os.makedirs(os.path.join(output_folder, "color"), exist_ok=True)
    img = (np.float32(36000) - depth).astype(np.float32)
    img[np.isinf(img)] = np.nan
    colmin = np.nanmin(np.where(np.isnan(img), np.inf, img), axis=0)     # per column, as the reference's loop
    bad = np.isnan(img)
    img[bad] = np.broadcast_to(colmin - 1, img.shape)[bad]
    plt.imsave(os.path.join(output_folder, "color", "%s_init.png" % name), img, format="png")
    plt.imsave(os.path.join(output_folder, "color", "%s_prob.png" % name), np.nan_to_num(prob).clip(0, 1), format="png")


def main(argv=None):
    a = parse_args(argv)
    if a.dataset != "cas_normal_eval":
        raise Exception("{}? Not implemented yet!".format(a.dataset))   # the other dataset classes are training sets
    if a.batch_size != 1:
        raise ValueError("--batch_size must be 1 (reference views are sharded over GPUs instead; predict.py:49)")
    if a.synthetic_items <= 0 and not a.data_folder:
        raise ValueError("--data_folder is required (or --synthetic_items N)")
    if not a.loadckpt and not (a.random_weights or a.synthetic_items > 0):
        # the reference fails at torch.load here (predict.py:105); never write products of an untrained net silently
        raise FileNotFoundError("--loadckpt is required (pass --random_weights to run on seeded random weights)")
    rank, world = sharding.init_from_env()
    # one process per GPU; more ranks than devices (tests on a one-GPU box) share the devices round-robin
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    # the dataset first: a bad folder fails before the model is built
    if a.synthetic_items > 0:
        ds = SyntheticBlock(a.synthetic_items, a.view_num, a.max_h, a.max_w, a.numdepth)
    else:
        from . import dataset

        ds = dataset.DeviceItems(dataset.MVSDataset(a.data_folder, "val", a.view_num, a.normalize, a))
    model = build_model(a.model, a.numdepth, [int(x) for x in a.ndepths.split(",")],
                        [float(x) for x in a.depth_inter_r.split(",")], share_cr=a.share_cr,
                        cr_base_chs=[int(x) for x in a.cr_base_chs.split(",")])
    if a.loadckpt:
        print("loading model {}".format(a.loadckpt))
        load_checkpoint(model, a.loadckpt)
    else:
        synthetic.fill_state_dict_(model.state_dict(), 0)
    model = model.cuda()
    st = {}
    cache_bytes = int(a.feature_cache_gb * (1 << 30)) if a.synthetic_items <= 0 else 0
    if a.fuse:
        from . import fuse, pipeline

        tm = {}
        checker = fuse.ConsistencyChecker(a.position_threshold, a.depth_threshold, a.normal_threshold, a.photometric_threshold)
        blocks = None
        if a.fuse_partition == "scene_blocks":
            from . import dataset as _ds

            blocks = _ds.read_scene_blocks(a.blocks_file or os.path.join(a.data_folder, "blocks.txt"))
        res = pipeline.predict_and_fuse(model, ds, a.output_folder, rank, world, checker=checker, fusion_num=a.fusion_num,
                                        min_geo_consist_num=a.geo_consist_num, filter_sources=bool(a.fuse_filter_sources),
                                        partition=a.partition, feature_cache_bytes=cache_bytes, timings=tm, display=_truthy(a.display),
                                        fuse_partition=a.fuse_partition, scene_blocks=blocks)
        pipeline.save_fused(res, a.fusion_output or os.path.join(a.output_folder, "fused"))
        print("rank %d/%d: %d views in %.2f s, all-gather of %.1f MB in %.2f ms (%s), fusion of its %d reference views %.2f s, "
              "%d vertices" % (rank, world, tm["views"], tm["predict_s"], tm["allgather_bytes"] / 1e6, tm["allgather_ms"], tm["backend"],
                               len(res), tm["fuse_s"], sum(int(r["points"]["xyz"].shape[0]) for r in res)))
        return [r["ref"] for r in res]
    names = predict_views(model, ds, a.output_folder, rank, world, display=_truthy(a.display),
                          feature_cache_bytes=cache_bytes, partition=a.partition, stats=st)
    acc = st.get("cache_hits", 0) + st.get("cache_misses", 0)
    print("rank %d/%d wrote %d views (%s partition): feature cache %d hits / %d lookups (%.0f %%), %.2f pyramids computed per view"
          % (rank, world, len(names), st.get("partition"), st.get("cache_hits", 0), acc,
             100.0 * st.get("cache_hits", 0) / acc if acc else 0.0, st.get("pyramids_per_view", 0.0)))
    return names


if __name__ == "__main__":
    main()
