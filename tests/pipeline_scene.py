"""Test fixture for tests/test_pipeline_gpu.py: a block whose depth maps are geometrically CONSISTENT, so that the fusion
step behind the all-gather has something to confirm (a random-weight network's maps agree nowhere: every mask would be empty).

`SceneViews` is a dataset in predict_views' item layout whose views look at one tilted plane (synthetic.make_fusion_scene);
`SceneModel` stands in for an Infer_* module: its forward returns the rendered depth / confidence map of the item's reference
view, found through the marker the dataset writes into depth_values.  Everything downstream -- predict_views' products, the
all-gather, the ownership rule, fuse_block, extract_points -- is the production code.

Run as a script it is one rank of a torch.distributed.run launch:
    python -m torch.distributed.run --nproc-per-node 2 tests/pipeline_scene.py <out_dir> <filter_sources 0|1> [views|scene_blocks]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from deep3d_aerial_amd import synthetic as S  # noqa: E402

H, W, N_VIEWS, FUSION_NUM = 96, 128, 7, 4


class SceneViews(object):
    def __init__(self, n=N_VIEWS, h=H, w=W, seed=5):
        ref, srcs = S.make_fusion_scene(h, w, n - 1, seed=seed, noise=0.002)
        self.views = [ref] + srcs
        rng = np.random.default_rng(seed + 1)
        for v in self.views:   # every view can be a reference view: each has a confidence map
            v.setdefault("confidence", rng.uniform(0.0, 1.0, (h, w)).astype(np.float32))
        self.h, self.w = h, w

    def __len__(self):
        return len(self.views)

    def view_records(self, fusion_num=10):
        n = len(self.views)
        return [{"name": "scene_%02d" % i, "src": ["scene_%02d" % ((i + k) % n) for k in range(1, min(n, 1 + fusion_num))],
                 "id": i + 1, "image": i} for i in range(n)]

    def __getitem__(self, idx):
        v = self.views[idx]
        cam = np.zeros((2, 4, 4), np.float32)
        cam[0] = v["E"]
        cam[1, :3, :3] = v["K"]
        cam[1, 3] = [400.0, 1.0, 384, 800.0]
        pm = {k: np.tile(np.eye(4, dtype=np.float32), (2, 1, 1)) for k in ("stage1", "stage2", "stage3")}
        name = "scene_%02d" % idx
        return {"imgs": np.zeros((2, 3, self.h, self.w), np.float32), "proj_matrices": pm,
                "depth_values": np.array([float(idx), 0.0], np.float32),   # the marker SceneModel reads
                "outcam": cam, "outlocation": [str(self.w), str(self.h), str(idx), name + ".png"], "ref_image_path": name + ".png"}


class SceneModel(torch.nn.Module):
    def __init__(self, scene):
        super().__init__()
        self.depth = [torch.from_numpy(v["depth"]) for v in scene.views]
        self.conf = [torch.from_numpy(v["confidence"]) for v in scene.views]

    def forward(self, imgs, proj_matrices, depth_values, image_keys=None):
        i = int(round(float(depth_values[0, 0])))
        dev = depth_values.device
        return {"depth": self.depth[i].to(dev)[None], "photometric_confidence": self.conf[i].to(dev)[None]}


def checker():
    from deep3d_aerial_amd import fuse

    return fuse.ConsistencyChecker(1.0, 0.01, 10.0, 0.2)


SCENE_BLOCKS = [{"scene_range": [-1e9, 1e9, -1e9, 1e9, -1e9, 1e9], "refs": [0, 1, 2]},
                {"scene_range": [-60.0, 40.0, -1e9, 1e9, -1e9, 1e9], "refs": [2, 3, 4]},          # (clips vertices in x; shares view 2)
                {"scene_range": [-1e9, 1e9, -1e9, 1e9, -1e9, 1e9], "refs": [4, 5, 6]}]


def main(out_dir, filter_sources, fuse_partition="views"):
    from deep3d_aerial_amd import pipeline, sharding

    rank, world = sharding.init_from_env()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    scene = SceneViews()
    tm = {}
    res = pipeline.predict_and_fuse(SceneModel(scene), scene, os.path.join(out_dir, "MVS"), rank, world, checker=checker(),
                                    fusion_num=FUSION_NUM, min_geo_consist_num=3, filter_sources=bool(filter_sources), timings=tm,
                                    fuse_partition=fuse_partition, scene_blocks=SCENE_BLOCKS if fuse_partition == "scene_blocks" else None)
    pipeline.save_fused(res, os.path.join(out_dir, "fused"))
    print("rank %d/%d fused %s: all-gather %.2f ms over %s" % (rank, world, [r["ref"] for r in res], tm["allgather_ms"], tm["backend"]))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else "views")
