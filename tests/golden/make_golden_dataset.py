#!/usr/bin/env python3
"""Golden vectors for SURVEY.md §8f row N3 (input side of a view), made by RUNNING THE REFERENCE's own
preprocess functions (mvs/mvs_cas/datasets/preprocess.py: scale_camera :19-30, crop_input :60-88, center_image
:92-117) in the build container:

    python tests/golden/make_golden_dataset.py

preprocess.py does `import cv2` at module level; OpenCV is not installed here and none of the three functions uses it,
so the name `cv2` is bound to an EMPTY module for the import (any use of it would raise).  Nothing is copied: the
module is imported in place from /root/reference.  The item builder itself (cas_normal_eval.py:94-182) needs
GDAL / imageio file readers and cannot run here; its matrix statements (:134-173) are checked by construction in
tests/test_dataset.py.  The .npz holds data only: seeded 8-bit images, cameras, and the reference's outputs.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("D3D_REFERENCE", "/root/reference")
sys.modules["cv2"] = types.ModuleType("cv2")
sys.path.insert(0, os.path.join(REF, "mvs", "mvs_cas", "datasets"))

import numpy as np  # noqa: E402
import preprocess as RP  # noqa: E402


def main():
    rng = np.random.default_rng(8101)
    out = {}
    cases = [(70, 106, 64, 96, "mean"), (64, 96, 64, 96, "mean"), (50, 75, 128, 160, "mean"), (90, 70, 64, 64, "standard"),
             (200, 333, 96, 128, "mean"), (40, 50, 32, 32, "vit")]
    for i, (h, w, max_h, max_w, mode) in enumerate(cases):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        img[:, :, 1] = (img[:, :, 1] // 3) + 40       # channels with different statistics
        cam = np.zeros((2, 4, 4), np.float32)
        cam[0] = np.eye(4)
        cam[0, :3, 3] = rng.standard_normal(3)
        cam[1, :3, :3] = [[1.1 * w, 0, w / 2 - 0.5], [0, 1.1 * w, h / 2 - 0.5], [0, 0, 1]]
        cam[1, 3] = [400, 1.5, 128, 592]
        if h <= max_h and w <= max_w and (h % 32 or w % 32):
            # crop_input pads nothing: for images smaller than the limits the window is rounded UP to a multiple of 32
            # and the slice simply clips -- the fixture keeps that case to pin the window arithmetic
            pass
        cropped, ccam = RP.crop_input(img.copy(), cam.copy(), max_h=max_h, max_w=max_w)
        centered = RP.center_image(cropped, mode=mode)
        scam = RP.scale_camera(ccam, scale=0.5)
        k = "c%d_" % i
        out.update({k + "img": img, k + "cam": cam, k + "max_hw": np.array([max_h, max_w]), k + "mode": np.array(mode),
                    k + "cropped": cropped, k + "crop_cam": ccam, k + "centered": centered.astype(np.float32),
                    k + "scaled_cam": scam})
        print(i, img.shape, "->", cropped.shape, centered.dtype, float(centered.mean()), float(centered.std()))
    out["n_cases"] = np.array(len(cases))
    path = os.path.join(HERE, "dataset_preprocess.npz")
    np.savez_compressed(path, **out)
    print("wrote %s %.1f KiB" % (os.path.basename(path), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
