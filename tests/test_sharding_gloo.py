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


def test_shard_views_is_a_partition():
    for n in (0, 1, 5, 8, 64, 67):
        for world in (1, 2, 3, 8):
            parts = [sharding.shard_views(n, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            for r, p in enumerate(parts):
                assert all(sharding.owner_of(i, world) == r for i in p)
    with pytest.raises(ValueError):
        sharding.shard_views(4, 2, 2)


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
