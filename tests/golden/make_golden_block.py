#!/usr/bin/env python3
"""Golden item tensors for the block reader (SURVEY.md 8b boundary B1), made by RUNNING THE
REFERENCE's own dataset class (mvs/mvs_cas/datasets/cas_normal_eval.py:10-182 with data_io.py:48-126 and
preprocess.py:19-117) on the synthetic block of tests/block_fixture.py:

    python tests/golden/make_golden_block.py

What is bound for the import, and why it does not change the arithmetic that is pinned:
  * `cv2`, `gdal`: absent from the image; imported at module level by data_io.py / preprocess.py.  Bound to modules
    whose only attribute is cv2.resize / cv2.INTER_LINEAR|INTER_NEAREST; `resize` accepts fx = fy = 1 only and returns
    the image unchanged (what OpenCV returns for a unit scale) -- any other use raises.  The fixture therefore pins
    resize_scale = 1, the value the pipeline runs with (mvs_dl.py:61-63 passes no --resize_scale).
  * `imageio`: absent; cas_normal_eval.py imports imread / imsave / imwrite by name and uses them only in read_depth
    (not on the inference path).  Bound to functions that raise.
  * `numpy.float`: create_cams (cas_normal_eval.py:65) uses the alias NumPy removed in 1.24; bound to `float`, which
    is what the alias was.
Nothing is copied: the modules are imported in place from /root/reference.  The .npz holds data only."""
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("D3D_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402

np.float = float


def _resize(image, dsize, fx=None, fy=None, interpolation=None):
    if dsize is not None or fx != 1 or fy != 1:
        raise RuntimeError("cv2 is not installed: only the unit-scale resize is available to the golden run")
    return image


def _absent(*a, **k):
    raise RuntimeError("imageio is not installed")


cv2 = types.ModuleType("cv2")
cv2.resize, cv2.INTER_LINEAR, cv2.INTER_NEAREST = _resize, 1, 0
sys.modules["cv2"] = cv2
sys.modules["gdal"] = types.ModuleType("gdal")
imageio = types.ModuleType("imageio")
imageio.imread = imageio.imsave = imageio.imwrite = _absent
sys.modules["imageio"] = imageio
sys.path.insert(0, os.path.join(REF, "mvs", "mvs_cas"))

import block_fixture as BF  # noqa: E402
from datasets.cas_normal_eval import MVSDataset  # noqa: E402  (the reference)


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        folder = BF.write_block(os.path.join(tmp, "block"))
        for normalize in ("mean", "standard"):
            ds = MVSDataset(folder, "val", BF.VIEW_NUM, normalize, BF.Args())
            out["n_%s" % normalize] = np.array(len(ds))
            for i in range(len(ds)):
                it = ds[i]
                k = "%s_%d_" % (normalize, i)
                out[k + "imgs"] = np.asarray(it["imgs"], np.float32)
                for st in ("stage1", "stage2", "stage3"):
                    out[k + "proj_" + st] = np.asarray(it["proj_matrices"][st])
                    out[k + "intri_" + st] = np.asarray(it["intri_matrices"][st])
                out[k + "depth_values"] = np.asarray(it["depth_values"])
                out[k + "outimage"] = np.asarray(it["outimage"])
                out[k + "outcam"] = np.asarray(it["outcam"])
                out[k + "outlocation"] = np.array(it["outlocation"])
                out[k + "ref_name"] = np.array(os.path.basename(it["ref_image_path"]))
                print(normalize, i, it["imgs"].shape, it["outlocation"], it["depth_values"], it["proj_matrices"]["stage3"].dtype)
    path = os.path.join(HERE, "block_items.npz")
    np.savez_compressed(path, **out)
    print("wrote %s %.1f KiB" % (os.path.basename(path), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
