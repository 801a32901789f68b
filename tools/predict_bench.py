"""End-to-end per-view time of the predict.py loop at 2752x1856, 5 views: model only, model +
asynchronous PFM products (PfmWriter: device flip, pinned D2H on a copy stream, writer thread), and model + the
reference's synchronous order (D2H, np.flipud, tofile before the next view starts; predict.py:146-183).
Usage: python tools/predict_bench.py [--model casmvsnet --views 6 --out /tmp/d3d_out]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import predict as P, synthetic as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="casmvsnet")
    ap.add_argument("--views", type=int, default=6)
    ap.add_argument("--h", type=int, default=1856)
    ap.add_argument("--w", type=int, default=2752)
    ap.add_argument("--out", default="/tmp/d3d_out")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    model = P.build_model(a.model, 384)
    S.fill_state_dict_(model.state_dict(), 1)
    model = model.cuda().eval()
    s = P.SyntheticBlock(1, 5, a.w, a.h, 384, seed=9)[0]
    imgs = torch.from_numpy(np.ascontiguousarray(s["imgs"]))[None].cuda()
    pm = {k: torch.from_numpy(np.ascontiguousarray(v))[None].cuda() for k, v in s["proj_matrices"].items()}
    dv = torch.from_numpy(np.ascontiguousarray(s["depth_values"]))[None].cuda()

    def run(mode):
        writer = None
        with torch.no_grad():
            model(imgs, pm, dv)
            torch.cuda.synchronize()
            t0 = time.time()
            for i in range(a.views):
                out = model(imgs, pm, dv)
                depth, prob = out["depth"].squeeze().contiguous(), out["photometric_confidence"].squeeze().contiguous()
                paths = [os.path.join(a.out, "v%d_init.pfm" % i), os.path.join(a.out, "v%d_prob.pfm" % i)]
                if mode == "async":
                    if writer is None:
                        writer = P.PfmWriter(depth.shape[0], depth.shape[1], 2)
                    writer.submit([depth, prob], paths)
                elif mode == "sync":
                    P.save_pfm(paths[0], np.float32(np.squeeze(depth.cpu().numpy())))
                    P.save_pfm(paths[1], np.float32(np.squeeze(prob.cpu().numpy())))
            if writer is not None:
                writer.close()
            torch.cuda.synchronize()
            return (time.time() - t0) / a.views * 1e3

    for mode in ("none", "async", "sync", "async", "sync"):
        print("%-6s products: %7.1f ms per view" % (mode, run(mode)), flush=True)

    # row N3: a strip whose reference views share images, items pre-built (decoded 8-bit images + cameras), whole loop
    # = upload, crop + normalise on the device, forward, asynchronous products
    strip = P.SyntheticStrip(a.views + 4, 5, a.w, a.h, 384, seed=3)
    items = [strip[i] for i in range(a.views)]
    for cache in (0, 64 << 30, 0, 64 << 30):
        P.predict_views(model, items[:1], a.out)
        torch.cuda.synchronize()
        t0 = time.time()
        P.predict_views(model, items, a.out, feature_cache_bytes=cache)
        torch.cuda.synchronize()
        print("strip loop, feature cache %2d GiB: %7.1f ms per view" % (cache >> 30, (time.time() - t0) / a.views * 1e3),
              flush=True)
    # the same strip in the reference's own item layout (host-normalised float "imgs", no keys): shared images are
    # recognised by content (fingerprint + exact comparison)
    from deep3d_aerial_amd import dataset as DS
    ref_items = []
    for it in items:
        it = dict(it)
        imgs = [DS.center_image(torch.from_numpy(im).cuda(), it["normalize"], w).cpu().numpy()
                for im, w in zip(it.pop("images_u8"), it.pop("crop_windows"))]
        it.pop("image_keys")
        it["imgs"] = np.stack(imgs)
        ref_items.append(it)
    for cache in (0, 64 << 30, 0, 64 << 30):
        P.predict_views(model, ref_items[:1], a.out)
        torch.cuda.synchronize()
        t0 = time.time()
        P.predict_views(model, ref_items, a.out, feature_cache_bytes=cache)
        torch.cuda.synchronize()
        print("reference item layout, content-matched cache %2d GiB: %7.1f ms per view" % (cache >> 30, (time.time() - t0) / a.views * 1e3),
              flush=True)
    # the reference's host-side normalisation of the same five images (preprocess.py:98-103), for scale
    t0 = time.time()
    for im in items[0]["images_u8"]:
        x = im[3:3 + a.w, 5:5 + a.h].astype(np.float32)  # (the strip's images are [max_h + 6, max_w + 10])
        var = np.var(x, axis=(0, 1), keepdims=True)
        mean = np.mean(x, axis=(0, 1), keepdims=True)
        _ = (x - mean) / (np.sqrt(var) + 0.00000001)
    print("host NumPy normalisation of one item's 5 images: %.1f ms" % ((time.time() - t0) * 1e3))


if __name__ == "__main__":
    main()
