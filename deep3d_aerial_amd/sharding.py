"""Multi-GPU execution of the plane-sweep path: one process per GPU, reference views sharded,
no collective on the data path.

The reference runs one GPU (nn.DataParallel at batch 1, predict.py:49,100) and hands depth maps
to fusion through PFM files (predict.py:179-183 -> fuse/fusion_3d_normal.py:433,448).  A
reference view (ref image + its sources from viewpair.txt) is an independent work item
(datasets/cas_normal_eval.py:94-182), so the views of a block are partitioned over the ranks and
every rank sweeps its own list.  The partition is CONTIGUOUS by default (rank r takes views
[r n / N, (r + 1) n / N)): consecutive reference views of a block share most of their source images
(viewpair.txt lists neighbours), so a rank's feature cache (dataset.FeatureCache) keeps hitting; dealt
round-robin, neighbouring views land on different ranks and at N = 8 the cache hits on almost nothing
(round 3's default; kept as policy="round_robin" for callers whose items are unrelated).
The only exchange step is OPTIONAL: an all-gather of
the per-view (depth, confidence) maps so that every rank holds its neighbours' maps for the
geometric consistency check (fuse/consistency_check_n.py:141-147).  It runs over
torch.distributed -- backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import os

import torch
import torch.distributed as dist


POLICIES = ("block", "round_robin")


def _check(rank, world_size, policy):
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    if policy not in POLICIES:
        raise ValueError("unknown partition %r (one of %s)" % (policy, ", ".join(POLICIES)))


def shard_views(n_views, rank, world_size, policy="block"):
    """Indices of the reference views rank `rank` sweeps, ascending.
    "block": the contiguous run [rank * n // N, (rank + 1) * n // N) -- sizes differ by at most one;
    "round_robin": i = rank (mod world_size)."""
    _check(rank, world_size, policy)
    if policy == "round_robin":
        return list(range(rank, n_views, world_size))
    return list(range(rank * n_views // world_size, (rank + 1) * n_views // world_size))


def owner_of(view_index, world_size, n_views=None, policy="block"):
    """Rank that sweeps view `view_index` (the block partition needs the number of views)."""
    _check(0, world_size, policy)
    if policy == "round_robin":
        return view_index % world_size
    if n_views is None:
        raise ValueError("owner_of: the block partition depends on n_views")
    if not (0 <= view_index < n_views):
        raise ValueError("view %d outside 0..%d" % (view_index, n_views - 1))
    # the r with r * n // N <= i < (r + 1) * n // N
    r = min(world_size - 1, (view_index * world_size + world_size - 1) // max(n_views, 1))
    while r > 0 and r * n_views // world_size > view_index:
        r -= 1
    while (r + 1) * n_views // world_size <= view_index:
        r += 1
    return r


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    (what torch.distributed.run exports).  Returns (rank, world_size)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            # RCCL needs one device per rank; ranks sharing a device (tests) and CPU runs rendezvous over gloo
            local = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
            backend = "nccl" if torch.cuda.is_available() and torch.cuda.device_count() >= local else "gloo"
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def all_gather_maps(local_maps, n_views, rank=None, world_size=None, policy="block"):
    """All-gather the per-view (depth, confidence) maps.

    local_maps: tensor [n_local, 2, H, W] holding this rank's views in shard_views() order.
    Returns [n_views, 2, H, W] ordered by global view index, identical on every rank.
    Ranks with fewer views are padded to the common count; one collective.
    """
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    mine = shard_views(n_views, rank, world_size, policy)
    if local_maps.shape[0] != len(mine):
        raise ValueError("rank %d holds %d maps, expected %d" % (rank, local_maps.shape[0], len(mine)))
    if world_size == 1:
        return local_maps
    per = (n_views + world_size - 1) // world_size
    padded = local_maps.new_zeros((per,) + tuple(local_maps.shape[1:]))
    padded[: local_maps.shape[0]] = local_maps
    if dist.get_backend() == "gloo" and padded.is_cuda:
        # ranks that share one card rendezvous over gloo (init_from_env): the maps travel through host memory
        host = padded.cpu()
        gathered = host.new_empty((world_size * per,) + tuple(host.shape[1:]))
        dist.all_gather_into_tensor(gathered, host.contiguous())
        gathered = gathered.to(local_maps.device)
    else:
        gathered = local_maps.new_empty((world_size * per,) + tuple(local_maps.shape[1:]))
        dist.all_gather_into_tensor(gathered, padded.contiguous())
    # gathered[r*per + j] is the j-th view of rank r's list
    if policy == "block" and n_views == world_size * per:
        return gathered   # equal contiguous blocks: rank-major order IS the global view order (no second 5 GB pass at config 5's size)
    out = local_maps.new_empty((n_views,) + tuple(local_maps.shape[1:]))
    for r in range(world_size):
        idx = shard_views(n_views, r, world_size, policy)
        out[idx] = gathered[r * per: r * per + len(idx)]
    return out


def run_sharded(process_view, n_views, rank=None, world_size=None, gather=False, policy="block"):
    """Sweep this rank's views with `process_view(i) -> tensor [2,H,W]` (depth, confidence).

    Returns {view index: tensor} for the local views, or -- with gather=True -- the
    [n_views,2,H,W] tensor every rank needs for fusion.
    """
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    mine = shard_views(n_views, rank, world_size, policy)
    results = {i: process_view(i) for i in mine}
    if not gather:
        return results
    if mine:
        local = torch.stack([results[i] for i in mine])
    else:
        raise ValueError("gather=True needs at least one view per rank (n_views >= world_size)")
    return all_gather_maps(local, n_views, rank, world_size, policy)
