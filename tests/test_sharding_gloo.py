"""Multi-process tests of the N > 1 path on CPU (gloo, world_size 2 and 3): view sharding is a
partition, per-view results do not depend on the number of ranks, and the optional all-gather
equals a host-side concatenation."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from deep3d_aerial_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("policy", sharding.POLICIES)
def test_shard_views_is_a_partition(policy):
    for n in (0, 1, 5, 8, 64, 67):
        for world in (1, 2, 3, 8):
            parts = [sharding.shard_views(n, r, world, policy) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            for r, p in enumerate(parts):
                assert p == sorted(p)
                assert all(sharding.owner_of(i, world, n, policy) == r for i in p)
                if policy == "block" and p:   # contiguous runs, in rank order
                    assert p == list(range(p[0], p[0] + len(p)))
    with pytest.raises(ValueError):
        sharding.shard_views(4, 2, 2, policy)
    with pytest.raises(ValueError):
        sharding.shard_views(4, 0, 2, "striped")


def _cache_run(strip, views, max_bytes=1 << 30):
    """The feature cache of one rank over its list of reference views: (hits, misses, pyramids computed)."""
    from deep3d_aerial_amd import dataset as DS

    cache = DS.FeatureCache(max_bytes)
    computed = []

    def feature_net(x):   # stands in for the image pyramid: what matters here is how often it runs
        computed.append(1)
        return {"stage1": x * 2.0}

    for i in views:
        keys = strip[i]["image_keys"]
        imgs = [None if k in cache else torch.full((1, 3, 2, 2), float(k[1])) for k in keys]
        feats = DS.extract_features(feature_net, imgs, keys, cache)
        assert [float(f["stage1"].flatten()[0]) for f in feats] == [2.0 * k[1] for k in keys]
    return cache.hits, cache.misses, len(computed)


@pytest.mark.parametrize("world", [2, 8])
def test_block_partition_keeps_the_feature_cache_hitting(world):
    """consecutive reference views of a block share their source images (viewpair.txt lists neighbours;
    predict.SyntheticStrip models it: view i uses images i .. i + V - 1).  Dealt round-robin over 8 ranks no rank ever sees
    an image twice; in contiguous blocks a rank's hit rate stays within 1 / views-per-rank of the single-GPU rate."""
    from deep3d_aerial_amd import predict as P

    n, V = 48, 5
    strip = P.SyntheticStrip(n, V, 64, 96, 64, seed=3)
    h1, m1, c1 = _cache_run(strip, range(n))
    rate1 = h1 / (h1 + m1)
    assert c1 == n and abs(rate1 - (1 - 1 / V)) < 1e-9          # every image featurised once
    per_rank = n // world
    total_block, total_rr = 0, 0
    for r in range(world):
        h, m, c = _cache_run(strip, sharding.shard_views(n, r, world))
        assert h / (h + m) >= rate1 - 1.0 / per_rank
        total_block += c
        h, m, c = _cache_run(strip, sharding.shard_views(n, r, world, "round_robin"))
        total_rr += c
        if world >= V:
            assert h == 0                                         # the round-3 default: the cache never hits
    assert total_block <= n + world * (V - 1)                     # only the seams between blocks are featurised twice
    assert total_rr >= total_block and (world < V or total_rr == n * V)


def _fake_view(i, h=6, w=5):
    """Deterministic stand-in for one reference view's (depth, confidence) maps."""
    g = torch.Generator().manual_seed(1000 + i)
    return torch.rand((2, h, w), generator=g) + i


def _worker(rank, world, port, n_views, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w = sharding.init_from_env("gloo")
    assert (r, w) == (rank, world)
    local = sharding.run_sharded(_fake_view, n_views)
    assert sorted(local) == sharding.shard_views(n_views, rank, world)
    full = sharding.run_sharded(_fake_view, n_views, gather=True)
    torch.save({"local": local, "full": full}, os.path.join(out_dir, "rank%d.pt" % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,n_views", [(2, 5), (2, 8), (3, 7)])
def test_sharded_run_and_all_gather(tmp_path, world, n_views):
    mp.spawn(_worker, args=(world, _free_port(), n_views, str(tmp_path)), nprocs=world, join=True)
    want = torch.stack([_fake_view(i) for i in range(n_views)])
    for rank in range(world):
        got = torch.load(os.path.join(str(tmp_path), "rank%d.pt" % rank))
        # per-view outputs are bitwise independent of the number of ranks
        for i, t in got["local"].items():
            assert torch.equal(t, want[i])
        # the all-gather equals the host concatenation, on every rank
        assert torch.equal(got["full"], want)


def test_single_process_degenerates():
    out = sharding.run_sharded(_fake_view, 4, rank=0, world_size=1, gather=True)
    assert torch.equal(out, torch.stack([_fake_view(i) for i in range(4)]))
