"""A tiny synthetic block in the reference's on-disk layout (viewpair.txt, images.txt, cameras.txt, image_path.txt +
PNG images), written from seeded generators.  Shared by tests/golden/make_golden_block.py (which runs the REFERENCE's
dataset class on it) and by the tests (which run this package's reader on the same files)."""
import os

import numpy as np

N_IMAGES, H0, W0 = 5, 100, 140
VIEW_NUM, NUM_DEPTH, MAX_H, MAX_W = 3, 64, 64, 96


def _rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def write_block(folder, seed=2024):
    """Five overlapping nadir images over a plane 500 m below, cameras in the file convention of images.txt
    ([Rwc | twc], x right / y up, looking along -z).  Returns the folder."""
    from PIL import Image

    os.makedirs(os.path.join(folder, "images"), exist_ok=True)
    rng = np.random.default_rng(seed)
    f = 1.3 * W0
    flip = np.diag([1.0, -1.0, -1.0])
    lines_img, lines_path = [], []
    for j in range(N_IMAGES):
        base = rng.integers(0, 256, (H0 // 4 + 1, W0 // 4 + 1, 3), dtype=np.uint8)
        img = np.repeat(np.repeat(base, 4, axis=0), 4, axis=1)[:H0, :W0]
        img = np.ascontiguousarray(img ^ rng.integers(0, 32, img.shape, dtype=np.uint8))
        path = os.path.join(folder, "images", "img_%02d.png" % j)
        Image.fromarray(img).save(path)
        # camera j: centre on a strip, small rotations; Tcw in x-right / y-down, then to the file convention
        Rcw = _rot(0.01 * np.sin(j + 1), 0.015 * np.cos(2 * j), 0.02 * (j - 2))
        C = np.array([35.0 * j, 4.0 * np.sin(1.7 * j), 1.5 * (j % 2)])
        Rwc_file = Rcw.T @ flip          # cas_normal_eval.create_cams multiplies by diag(1,-1,-1) again
        vals = list(Rwc_file.reshape(-1)) + list(C) + [430.0 + 3 * j, 590.0 - 2 * j]
        lines_img.append("%d %d %s %s" % (j, 1 + (j % 2), " ".join(repr(float(v)) for v in vals), "img_%02d.png" % j))
        lines_path.append("%d img_%02d %s" % (j, j, path))
    with open(os.path.join(folder, "cameras.txt"), "w") as fh:
        fh.write("# camera_id width height pixelsize fx fy x0 y0 k1 k2 k3 p1 p2\n")
        fh.write("1 %d %d 0.005 %r %r %r %r 0 0 0 0 0\n" % (W0, H0, f, f, (W0 - 1) / 2.0, (H0 - 1) / 2.0))
        fh.write("2 %d %d 0.005 %r %r %r %r 0 0 0 0 0\n" % (W0, H0, 1.01 * f, 0.99 * f, W0 / 2.0 - 1.25, H0 / 2.0 + 0.75))
    with open(os.path.join(folder, "images.txt"), "w") as fh:
        fh.write("# image_id camera_id R(9) C(3) depth_min depth_max name\n\n")
        fh.write("\n".join(lines_img) + "\n")
    with open(os.path.join(folder, "image_path.txt"), "w") as fh:
        fh.write("%d\n" % N_IMAGES + "\n".join(lines_path) + "\n")
    with open(os.path.join(folder, "viewpair.txt"), "w") as fh:
        # views 0..3 list 2-4 sources with scores; view 4 has none (dropped); view 3 has one (padded)
        fh.write("5\n0\n3 1 0.9 2 0.8 3 0.5\n1\n2 0 0.9 2 0.7\n2\n4 1 0.9 3 0.9 0 0.6 4 0.4\n3\n1 4 0.8\n4\n0\n")
    return folder


class Args(object):
    """The argparse fields the dataset reads (predict.py:38-48 defaults, sized for the fixture)."""
    min_interval, interval_scale, numdepth = 0.1, 1.0, NUM_DEPTH
    resize_scale, sample_scale, max_h, max_w = 1.0, 1.0, MAX_H, MAX_W
